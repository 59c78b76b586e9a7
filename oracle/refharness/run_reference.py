"""Drive the real SpliSER v0.1.8 reference in the build container (TEST INFRASTRUCTURE).

/root/reference is read-only and exists only in the build container, never on the GPU box.
This module is used by tests/golden/make_golden.py (to produce the committed fixtures) and by
container-only tests that compare the C/numpy restatement in oracle/ with the reference on
randomised inputs.  Nothing in the product imports it.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_DIR = os.environ.get("SPLISER_REFERENCE_DIR", "/root/reference")
REFERENCE_SCRIPT = os.path.join(REFERENCE_DIR, "SpliSER_v0_1_8.py")


def reference_available():
    return os.path.isfile(REFERENCE_SCRIPT)


def _env():
    env = dict(os.environ)
    env["PATH"] = HERE + os.pathsep + env.get("PATH", "")
    env["PYTHONPATH"] = HERE + os.pathsep + env.get("PYTHONPATH", "")
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    return env


def run_cli(args, inprocess=True, cwd=None, timeout=3600, dump_json=None):
    """Run ``SpliSER_v0_1_8.py <args>``; returns (returncode, stdout+stderr text)."""
    env = _env()
    if dump_json:
        if not inprocess:
            raise ValueError("dump_json needs the in-process replay")
        env["SPLISER_REF_DUMP"] = dump_json
    if inprocess:
        cmd = [sys.executable, os.path.join(HERE, "ref_inproc.py"), REFERENCE_SCRIPT] + list(args)
    else:
        cmd = [sys.executable, REFERENCE_SCRIPT] + list(args)
    res = subprocess.run(cmd, env=env, cwd=cwd, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=timeout)
    return res.returncode, res.stdout.decode("utf-8", "replace")


def process_args(sam, bed, out_prefix, gff=None, chrom=None, gene=None, max_intron=None,
                 stranded=None, cryptic=False):
    args = ["process", "-B", sam, "-b", bed, "-o", out_prefix]
    if gff is not None:
        args += ["-A", gff]
    if chrom is not None:
        args += ["-c", chrom]
    if gene is not None:
        args += ["-g", gene]
    if max_intron is not None:
        args += ["-m", str(max_intron)]
    if stranded:
        args += ["--isStranded", "-s", stranded]
    if cryptic:
        args += ["--beta2Cryptic"]
    return args


def run_process(sam, bed, out_prefix, inprocess=True, dump_json=None, **kw):
    rc, log = run_cli(process_args(sam, bed, out_prefix, **kw), inprocess=inprocess, dump_json=dump_json)
    if rc != 0:
        raise RuntimeError("reference process failed (%d):\n%s" % (rc, log))
    with open(out_prefix + ".SpliSER.tsv", "r") as handle:
        return handle.read(), log
