import os
import sys

from .cli import main

rc = main()
# Everything the command writes is written and closed when main() returns.  What is left -- the alignment file's mapping (14 GB
# take 0.07 s to unmap), tens of GB of device memory, the HIP runtime's own teardown: 0.16-0.2 s for a human-scale sample -- the
# kernel takes back faster than the process can hand it back: leave at once (SPL_NO_FAST_EXIT=1 for the ordinary way out).
sys.stdout.flush()
sys.stderr.flush()
if os.environ.get("SPL_NO_FAST_EXIT"):
    sys.exit(rc)
os._exit(rc or 0)
