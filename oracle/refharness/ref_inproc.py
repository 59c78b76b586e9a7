"""Run the *unmodified* reference script in this interpreter with the per-site
``samtools view`` child process answered from memory (TEST INFRASTRUCTURE, container only).

Usage:  python ref_inproc.py <reference_script.py> <reference CLI arguments...>

Only ``subprocess.Popen`` is replaced (with an object exposing ``.stdout`` as an iterable of
byte lines, which is all SpliSER_v0_1_8.py:422-427 touches); no reference code is edited.
The subprocess-per-site path (fake ``samtools`` on PATH) and this replay give identical
outputs (checked by make_golden.py); this one merely avoids ~30 ms of interpreter start-up
per splice site so that randomised cases with thousands of sites finish in seconds.
"""
import os
import runpy
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import samshim  # noqa: E402

_cache = {}


class _Replay(object):
    def __init__(self, argv, stdout=None, **kwargs):
        if list(argv[:2]) != ["samtools", "view"]:
            raise RuntimeError("unexpected child process: %r" % (argv,))
        path, region = argv[2], argv[3]
        if path not in _cache:
            _cache[path] = samshim.SamIndex(path)
        self.stdout = iter(_cache[path].query(region))


def main():
    script = sys.argv[1]
    sys.argv = [script] + sys.argv[2:]
    sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
    subprocess.Popen = _Replay
    ns = runpy.run_path(script, run_name="__main__")
    dump = os.environ.get("SPLISER_REF_DUMP")
    if dump and sys.argv[1] == "process":
        _dump_sites(ns, dump)


def _dump_sites(ns, path):
    """Full-precision per-site state of the reference after ``process`` (json floats are repr-exact)."""
    import json
    rows = []
    for ci, chrom in enumerate(ns["chrom_index"]):
        for site in ns["site2D_array"][ci]:
            rows.append({
                "chrom": chrom, "pos": site.getPos(), "strand": site.getStrand(), "gene": site.getGeneName(),
                "alpha": site.getAlphaCount(0), "beta1": site.getBeta1Count(0),
                "beta2Simple": site.getBeta2SimpleCount(0), "beta2Cryptic": site.getBeta2CrypticCount(0),
                "beta2Weighted": float(site.getBeta2WeightedCount(0)), "sse": float(site.getSSE(0)),
                "partners": [[int(k), int(v)] for k, v in site.getPartnerCount(0).items()],
                "competitors": [int(c) for c in site.getCompetitorPos()],
                "double": [[int(k), int(v[0])] for k, v in site.getPartnerBeta2DoubleCounts().items()],
            })
    with open(path, "w") as fh:
        json.dump(rows, fh)


if __name__ == "__main__":
    main()
