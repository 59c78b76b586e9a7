import os
import sys

_stamps = bool(os.environ.get("SPL_CLI_STAMPS"))     # (diagnostic: wall-clock stamps on stderr, for whoever started this process and wants to
if _stamps:                                           #  know what lies before main() and behind it -- tools/cli_cold.py --walls)
    import time
    sys.stderr.write("[cli stamp] __main__ %.6f\n" % time.time())

from .cli import main  # noqa: E402

if _stamps:
    sys.stderr.write("[cli stamp] imported %.6f\n" % time.time())
rc = main()
if _stamps:
    sys.stderr.write("[cli stamp] main returned %.6f\n" % time.time())
# Everything the command writes is written and closed when main() returns.  What is left -- the alignment file's mapping (14 GB
# take 0.07 s to unmap), tens of GB of device memory, the HIP runtime's own teardown: 0.16-0.2 s for a human-scale sample -- the
# kernel takes back faster than the process can hand it back: leave at once.  NOT when something rides along that writes its
# output at exit -- a profiler's or tracer's preloaded library (rocprofv3, LD_PRELOAD tools), coverage -- or when asked not to
# (SPL_NO_FAST_EXIT=1): then the ordinary way out, with atexit handlers, static destructors and the closing thread joined.
sys.stdout.flush()
sys.stderr.flush()
_rides_along = any(os.environ.get(k) for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_REGISTER_FORCE_LOAD",
                                               "COVERAGE_PROCESS_START", "COV_CORE_SOURCE")) or "coverage" in sys.modules or sys.gettrace() is not None
if os.environ.get("SPL_NO_FAST_EXIT") or _rides_along:
    from .process import wait_deferred_close
    wait_deferred_close()
    sys.exit(rc)
if _stamps:
    sys.stderr.write("[cli stamp] leaving %.6f\n" % time.time())
    sys.stderr.flush()
os._exit(rc or 0)
