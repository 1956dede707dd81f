"""prebuilds the timing-instrumented variant libraries of tools/time_b1.py / time_gather.py / time_ccl.py into variants/ (git-ignored, travels
with gpurun) so that a GPU session spends no time in hipcc"""
import importlib.util, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("mot_build", os.path.join(ROOT, "3d-lidar-multi-object-tracking_amd", "build.py"))
B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
os.makedirs(os.path.join(ROOT, "variants"), exist_ok=True)
for name, flags in (("dbg_libmot_b1t.so", ["-DMOT_DBG_B1_TIMING", "-DMOT_DBG_B1B_TIMING"]), ("dbg_libmot_timing.so", ["-DMOT_DBG_TIMING"]), ("dbg_libmot_cclt.so", ["-DMOT_DBG_CCL_TIMING"]), ("dbg_libmot_rect.so", ["-DMOT_DBG_RECT_TIMING"])):
    print(B.build(extra_flags=flags, out=os.path.join(ROOT, "variants", name)))
