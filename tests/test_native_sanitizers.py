"""AddressSanitizer + UBSan build of the host-side BAM reader/writer (GPU sanitizers are not available on the pool;
this is the CPU build the task allows)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bam_reader_writer_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "bam_asan")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-pthread",
           os.path.join(ROOT, "tests", "native", "bam_asan_driver.cpp"), os.path.join(ROOT, "spliser_amd", "csrc", "bam_reader.cpp"),
           "-o", exe, "-lz", "-ldl"]
    subprocess.check_call(cmd)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([exe, str(tmp_path), "200000"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = out.stdout.decode("utf-8", "replace")
    assert out.returncode == 0, text
    assert "ERROR: AddressSanitizer" not in text and "runtime error" not in text, text
    assert text.startswith("ok:")
