"""gpu_tiers.py with the option string taken from AVK_OPTS (for tools/timeline_run.sh, which passes no arguments through)."""
import os, runpy, sys
sys.argv = [sys.argv[0], os.environ.get("AVK_OPTS", "")]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_tiers.py"), run_name="__main__")
