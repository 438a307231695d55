"""Private build of librva with -DRVA_EXPERIMENTS (timing-only experiment kernels, variants 96.. -- 90 / 91 when round 4's A/Bs were recorded, before the stride-2 kernels took 86-93) into tools/_dbg/librva_exp.so
(git-ignored, travels with gpurun).  Use:  RVA_LIB_PATH=tools/_dbg/librva_exp.so python tools/sweep_run.py ... 56 96"""
import subprocess, sys
from pathlib import Path
sys.path.insert(0, ".")
from realtime_video_analytics_32streams_amd import _native as N
out = Path("tools/_dbg/librva_exp.so")
out.parent.mkdir(exist_ok=True)
subprocess.run(["hipcc", *N.HIPCC_FLAGS, "-DRVA_EXPERIMENTS", f"-I{N.ROOT / 'include'}", "-o", str(out), *[str(N.CSRC / s) for s in N.SOURCES], "-ldl"], check=True)
print(out)
