"""One module per IR layer type; the module name is the layer type (reference inference_engine.py:28-43)."""

# Every plugin of this package takes and returns device-resident tensors and enqueues its work on the current
# compute stream, so the engine may fork independent branches of a graph onto separate streams.
DEVICE_STREAMS = True

# The kernels of this package compute in fp32: an FP16 IR is loaded with its constants upcast and its ports declared FP32
# (IENetwork.promote_fp16).
COMPUTE_FP32 = True

# ... unless it is read with fp16_as_fp32=False: the tensors stay fp32 in HBM, but Convolution and MatMul then round their
# operands to fp16 and run on the f16 matrix-core instructions with fp32 accumulation (node['_f16_mfma'], set by the engine).
F16_MFMA = True
