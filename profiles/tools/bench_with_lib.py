import os, sys, runpy
sys.path.insert(0, os.path.join(os.getcwd(), "send-slam_amd"))
from send_slam_amd import binding
if os.environ.get("SENDSLAM_LIB"): binding.LIB_PATH = os.environ["SENDSLAM_LIB"]
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path("bench.py", run_name="__main__")
