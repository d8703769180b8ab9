"""What does a FAILED stream capture leave behind on this ROCm build, and what clears it?  (round 6: after an abandoned capture the next launch of the library
reported hipErrorStreamCaptureInvalidated.)  python scripts/capture_error_probe.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aigv_assessor_amd import native
lib = native.load()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipGetErrorString.restype = ctypes.c_char_p
hip.hipStreamEndCapture.restype = ctypes.c_int
hip.hipStreamEndCapture.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
hip.hipStreamIsCapturing.restype = ctypes.c_int
hip.hipStreamIsCapturing.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]


def end_capture():
    st = torch.cuda.graph.default_capture_stream.cuda_stream
    status = ctypes.c_int(-1)
    rc0 = hip.hipStreamIsCapturing(st, ctypes.byref(status))
    g = ctypes.c_void_p()
    rc = hip.hipStreamEndCapture(st, ctypes.byref(g))
    status2 = ctypes.c_int(-1)
    hip.hipStreamIsCapturing(st, ctypes.byref(status2))
    return f"isCapturing rc {rc0} status {status.value}; hipStreamEndCapture rc {rc} ({hip.hipGetErrorString(rc).decode()}); status after {status2.value}; lastError {hip.hipGetLastError()}"
x = torch.ones(1024, device="cuda")
y = torch.empty(1024, dtype=torch.bfloat16, device="cuda")
w = torch.ones(1024, dtype=torch.bfloat16, device="cuda")


def native_launch():
    return lib.aigv_op_rmsnorm(y.data_ptr(), 1024, w.data_ptr(), y.data_ptr(), 1024, 1, 1024, ctypes.c_float(1e-5), None, native.stream_ptr())


def torch_launch():
    try:
        (x + 1).sum().item()
        return "ok"
    except Exception as e:
        return f"{type(e).__name__}: {str(e).splitlines()[0]}"


KEEP = []


def fail_capture():
    g = torch.cuda.CUDAGraph()
    KEEP.append(g)                      # (destroying the graph object of a failed capture aborts the process on this torch build: leak it)
    try:
        with torch.cuda.graph(g, capture_error_mode="relaxed"):
            z = x + 1
            torch.cuda.current_stream().synchronize()
    except Exception as e:
        return f"{type(e).__name__}: {str(e).splitlines()[0]}"
    return "captured?!"


cap = torch.cuda.graphs.graph.default_capture_stream if hasattr(torch.cuda.graphs.graph, "default_capture_stream") else None
FIXES = {
    "nothing": lambda: None,
    "hipGetLastError x4": lambda: [hip.hipGetLastError() for _ in range(4)],
    "hipDeviceSynchronize + hipGetLastError": lambda: (hip.hipDeviceSynchronize(), hip.hipGetLastError()),
    "hipStreamEndCapture(capture stream) + hipGetLastError": end_capture,
    "hipThreadExchangeStreamCaptureMode(global) + hipGetLastError": lambda: (hip.hipThreadExchangeStreamCaptureMode(ctypes.byref(ctypes.c_int(0))), hip.hipGetLastError()),
    "a clean capture cycle on the same stream": lambda: clean_cycle(),
}


def clean_cycle():
    g = torch.cuda.CUDAGraph()
    KEEP.append(g)
    with torch.cuda.graph(g, capture_error_mode="relaxed"):
        z = x + 1
    return "clean capture ok"


name = sys.argv[1]
print("before: native", native_launch(), "torch", torch_launch())
print(f"--- failed capture: {fail_capture()}")
try:
    print("   fix returned", FIXES[name]())
except Exception as e:
    print("   fix raised", type(e).__name__, str(e).splitlines()[0])
r = native_launch()
print(f"after '{name}': native launch rc {r} ({lib.aigv_last_error(None).decode()[-90:] if r else 'ok'}); second native launch rc {native_launch()}; torch: {torch_launch()}", flush=True)
os._exit(0)
