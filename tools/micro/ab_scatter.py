import sys, runpy, os
sys.path[:0] = ["/root/repo", "/root/repo/construction-clip_amd"]
import cclip_hip.ops as o
o.SCATTER_DETERMINISTIC = os.environ.get("DET", "1") == "1"
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-extras", "--steps", "12", "--warmup", "3"]
runpy.run_path("/root/repo/bench.py", run_name="__main__")
