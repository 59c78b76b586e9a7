"""Command line of the MI355X build: the sub-commands, flags and argparse dests of SpliSER v0.1.8
(SpliSER_v0_1_8.py:1295-1361), so existing pipelines only swap the script name.

Extra flags (all optional, none changes results): ``--gpus`` / ``--devices`` to shard chromosomes over
several MI355X of one node, ``--threads`` for the BAM decode pool.  Extra sub-command: ``junctions`` writes the BED12
junction file ``process -b`` wants from the BAM itself (the reference leaves that to regtools).
"""
import argparse
import sys
import timeit

VERSION = "v0.1.8-mi355x"


def build_parser():
    parser = argparse.ArgumentParser(description="SpliSER - Splice Site Strength Estimates from RNA-seq (MI355X build)")
    sub = parser.add_subparsers(dest="command")
    p = sub.add_parser("process")
    p.add_argument("-B", "--BAMFile", dest="inBAM", required=True, help="The mapped RNA-seq file in BAM format")
    p.add_argument("-b", "--bedFile", dest="inBed", required=True, help="The Tophat-style splice junction bed file")
    p.add_argument("-o", "--outputPath", dest="outputPath", required=True,
                   help="Absolute path, including file prefix where the .SpliSER.tsv file is written")
    p.add_argument("-A", "--annotationFile", dest="annotationFile", required=False,
                   help="optional: gff3 or gtf file matching the reference genome used for alignment")
    p.add_argument("-t", "--annotationType", dest="aType", nargs="?", default="gene", type=str, required=False,
                   help="optional: the feature to be extracted from the annotation file - default: gene")
    p.add_argument("-c", "--chromosome", dest="qChrom", nargs="?", default="All", type=str, required=False,
                   help="optional: limit SpliSER to one chromosome/scaffold - default: All")
    p.add_argument("-g", "--gene", dest="qGene", nargs="?", default="All", type=str, required=False,
                   help="optional: limit SpliSER to splice sites falling in a single locus "
                        "(requires --chromosome, --annotationFile and --maxIntronSize)")
    p.add_argument("-m", "--maxIntronSize", dest="maxIntronSize", nargs="?", default=0, type=int, required=False,
                   help="optional: required with --gene, the max intron size used in aligning the bam file")
    p.add_argument("--isStranded", dest="isStranded", default=False, action="store_true")
    p.add_argument("-s", "--strandedType", dest="strandedType", nargs="?", type=str, required=False,
                   help='optional: strand specificity of the library, "rf" (first-strand) or "fr" (second-strand)')
    p.add_argument("--beta2Cryptic", dest="isbeta2Cryptic", default=False, action="store_true",
                   help="optional: weight the utilisation of competing splice sites into SSE (legacy)")
    p.add_argument("--checkJunctions", dest="checkJunctions", default=False, action="store_true",
                   help="(this build only) also count every junction in the BAM on the GPU and write <outputPath>.junctionCheck.tsv: "
                        "BED alpha against reads in the BAM; changes no result")
    p.add_argument("--gpuDecode", dest="gpuDecode", default=None, action="store_true",
                   help="(this build only) inflate the BAM's BGZF blocks and extract its records on the GPU whatever the file "
                        "looks like (the default with one GPU; with several, only files that compress like real libraries go "
                        "that way); changes no result")
    p.add_argument("--hostDecode", dest="gpuDecode", action="store_false",
                   help="(this build only) decode the BAM on host threads")
    p.add_argument("--keepReads", dest="keepReads", default=False, action="store_true",
                   help="(this build only) also write <outputPath>.SpliSER.reads (flag, POS, CIGAR of every read): combine takes it "
                        "instead of decoding the BAM again while it is still that BAM's")
    _engine_flags(p)
    c = sub.add_parser("combine")
    c.add_argument("-S", "--samplesFile", dest="samplesFile", required=True,
                   help="three-column .tsv: sample name, path of its .SpliSER.tsv, path of its BAM")
    c.add_argument("-o", "--outputPath", dest="outputPath", required=True, help="output prefix (.combined.tsv is appended)")
    c.add_argument("-g", "--gene", dest="qGene", nargs="?", default="All", type=str, required=False)
    c.add_argument("--isStranded", dest="isStranded", default=False, action="store_true")
    c.add_argument("-s", "--strandedType", dest="strandedType", nargs="?", default="fr", type=str, required=False)
    c.add_argument("--beta2Cryptic", dest="isbeta2Cryptic", default=False, action="store_true")
    _engine_flags(c)
    h = sub.add_parser("combineShallow")
    h.add_argument("-S", "--samplesFile", dest="samplesFile", required=True)
    h.add_argument("-g", "--gene", dest="qGene", nargs="?", default="All", type=str, required=False)
    h.add_argument("-o", "--outputPath", dest="outputPath", required=True)
    h.add_argument("--isStranded", dest="isStranded", default=False, action="store_true")
    h.add_argument("-m", "--minSamples", dest="minSamples", required=False, nargs="?", default=0, type=int)
    h.add_argument("-r", "--minReads", dest="minReads", required=False, nargs="?", default=10, type=int)
    h.add_argument("-e", "--minSSE", dest="minSSE", required=False, nargs="?", default=0.00, type=float)
    h.add_argument("-s", "--strandedType", dest="strandedType", nargs="?", type=str, required=False)
    h.add_argument("--beta2Cryptic", dest="isbeta2Cryptic", default=False, action="store_true")
    _engine_flags(h)
    o = sub.add_parser("output")
    o.add_argument("-S", "--samplesFile", dest="samplesFile", required=True)
    o.add_argument("-C", "--combinedFile", dest="combinedFile", required=True)
    o.add_argument("-t", "--outputType", dest="outputType", required=True, help="DiffSpliSER or GWAS")
    o.add_argument("-o", "--outputPath", dest="outputPath", required=True)
    o.add_argument("-r", "--minReads", dest="minReads", required=False, nargs="?", default=10, type=int)
    o.add_argument("-g", "--gene", dest="qGene", required=False, nargs="?", default="All", type=str)
    o.add_argument("-m", "--minSamples", dest="minSamples", required=False, nargs="?", default=50, type=int)
    j = sub.add_parser("junctions", help="(this build only) BED12 junction file for `process -b`, derived from the BAM on the GPU")
    j.add_argument("-B", "--BAMFile", dest="inBAM", required=True)
    j.add_argument("-o", "--outputPath", dest="outputPath", required=True, help="path of the BED12 file to write")
    j.add_argument("-c", "--chromosome", dest="qChrom", nargs="?", default="All", type=str, required=False)
    j.add_argument("--isStranded", dest="isStranded", default=False, action="store_true")
    j.add_argument("-s", "--strandedType", dest="strandedType", nargs="?", type=str, required=False)
    j.add_argument("-a", "--minAnchor", dest="minAnchor", type=int, default=8, help="both anchors of a read must be this long (regtools -a)")
    j.add_argument("-m", "--minIntron", dest="minIntron", type=int, default=70, help="regtools -m")
    j.add_argument("-M", "--maxIntron", dest="maxIntron", type=int, default=500000, help="regtools -M; 0 = no limit")
    _engine_flags(j)
    return parser


def _engine_flags(p):
    p.add_argument("--gpus", dest="gpus", type=int, default=1, help="number of MI355X devices to shard chromosomes over")
    p.add_argument("--devices", dest="devices", type=str, default=None, help="explicit device list, e.g. 0,2,3")
    p.add_argument("--threads", dest="threads", type=int, default=0, help="host threads for BAM decode (0 = all cores)")


def main(argv=None):
    print("\nSpliSER " + VERSION + " (MI355X / gfx950 build of SpliSER v0.1.8, SKB LAB)\n")
    start = timeit.default_timer()
    parser = build_parser()
    kwargs = vars(parser.parse_args(argv))
    command = kwargs.pop("command")
    if command is None:
        parser.error("a sub-command is required")
    gpus = kwargs.pop("gpus", 1)
    devices = kwargs.pop("devices", None)
    devices = tuple(int(d) for d in devices.split(",")) if devices else tuple(range(max(1, gpus)))
    threads = kwargs.pop("threads", 0)
    # same validation rules as SpliSER_v0_1_8.py:1350-1355
    if command == "process" and kwargs.get("qGene") != "All" and (kwargs.get("annotationFile") is None or kwargs.get("maxIntronSize") is None):
        print(kwargs.get("qGene"))
        print(kwargs.get("annotationFile"))
        parser.error("--gene requires --annotationFile and --maxIntronSize")
    elif command in ("process", "combine", "combineShallow", "junctions") and kwargs.get("isStranded") is True and kwargs.get("strandedType") is None:
        parser.error("--isStranded requires parameter --strandedType/-s as fr or rf")
    if command == "process":
        from .process import process
        if __import__("os").environ.get("SPL_PROCESS_TIMING"):
            sys.stderr.write("[cli] modules of `process` imported %.4f s after main() began\n" % (timeit.default_timer() - start))
        process(devices=devices, threads=threads, **kwargs)
        if kwargs.get("keepReads"):      # (the kept reads go out on the thread that closes the alignment file: the command is done when they are)
            from .process import wait_deferred_close
            wait_deferred_close()
    elif command == "combine":
        from .combine import combine
        combine(devices=devices, threads=threads, **kwargs)
    elif command == "combineShallow":
        from .combine import combineShallow
        combineShallow(devices=devices, threads=threads, **kwargs)
    elif command == "output":
        from .output import output
        output(**kwargs)
    elif command == "junctions":
        from .junctions import junctions
        junctions(devices=devices, threads=threads, **kwargs)
    else:
        parser.error("sub-command %r is not part of this build yet" % command)
    print("Total runtime (s): \t" + str(timeit.default_timer() - start))
    return 0


if __name__ == "__main__":
    sys.exit(main())
