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


@pytest.mark.parametrize("batch_blocks", ["1", "32"])
def test_bam_decoder_under_thread_sanitizer(tmp_path, batch_blocks):
    """The decoder hands batches from its workers to the committing thread without a lock (state word per ring slot, frontier
    counter, block directory growing beside it): ThreadSanitizer on the same driver, small batches included (every batch then
    starts inside a record and the ring turns over thousands of times)."""
    exe = str(tmp_path / "bam_tsan")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread",
           os.path.join(ROOT, "tests", "native", "bam_asan_driver.cpp"), os.path.join(ROOT, "spliser_amd", "csrc", "bam_reader.cpp"),
           "-o", exe, "-lz", "-ldl"]
    built = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if built.returncode != 0:
        pytest.skip("no ThreadSanitizer runtime in this image: " + built.stdout.decode("utf-8", "replace")[-200:])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", SPL_BAM_BATCH_BLOCKS=batch_blocks)
    out = subprocess.run([exe, str(tmp_path), "60000"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    text = out.stdout.decode("utf-8", "replace")
    assert out.returncode == 0 and "ThreadSanitizer" not in text, text[-3000:]
    assert text.startswith("ok:")
