"""Run bench.py with a watchdog that dumps every thread's Python stack after PROBE_S seconds and exits:
python3 tools/hang_probe.py [bench arguments]   (also under torch.distributed.run)."""
import faulthandler
import os
import runpy
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
faulthandler.dump_traceback_later(int(os.environ.get("PROBE_S", "90")), exit=True)
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[1:]
sys.path.insert(0, root)
runpy.run_path(sys.argv[0], run_name="__main__")
