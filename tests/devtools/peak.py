"""developer script: measured FP64 matrix peak of the device (hipsdp_mfma_peak)"""
import ctypes as C, importlib.util, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("hb", os.path.join(ROOT, "scip-sdp_amd", "binding.py")); hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
tf, g = C.c_double(0), C.c_double(0)
for ms in (5.0, 20.0, 50.0):
    print(ms, hb.ulib().hipsdp_mfma_peak(0, C.c_double(ms), C.byref(tf), C.byref(g)), "%.1f TFLOP/s at %.2f GHz" % (tf.value, g.value))
