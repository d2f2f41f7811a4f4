# DetectionOutput -- HIP plugin.  Replaces kernel_DetectionOutput_naive (reference
# op_plugins/DetectionOutput.py:163-259; attribute parsing :272-305): best class per prior, confidence screen,
# box decoding, the reference's all-pairs suppression, clipping and the score-ordered record list, one workgroup
# per image (csrc/pvhip_detect.hip).  The reference asserts N == 1; here images are independent and image b's
# records are rows [b * records, (b + 1) * records) of the (1, 1, N * records, 7) output, column 0 being the record
# index inside the image as in the reference.
import ctypes

from .. import common_def
from .. import device as dev


# A pass made of such nodes can be recorded into a hipGraph on ONE stream (Executable_Network.infer does so by itself for
# device-resident inputs; the whole SSD IR was tried: scripts/repro_capture.py).
GRAPH_CAPTURE_SAFE = True

def name():
    print('DetectionOutput')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'hip', debug: bool = False):
    if debug:
        print(node)
    common_def.validate_inputs(node, inputs)
    d = node['data']
    num_classes = int(d['num_classes'])
    top_k = int(d['top_k']) if 'top_k' in d else -1
    variance_encoded = common_def.string_to_boolean(d['variance_encoded_in_target']) if 'variance_encoded_in_target' in d else False
    keep_top_k = common_def.string_to_tuple(d['keep_top_k'])
    code_type = d['code_type'] if 'code_type' in d else 'caffe.PriorBoxParameter.CORNER'
    share_location = common_def.string_to_boolean(d['share_location']) if 'share_location' in d else True
    nms_threshold = float(d['nms_threshold'])
    confidence_threshold = float(d['confidence_threshold']) if 'confidence_threshold' in d else 0
    clip_after = common_def.string_to_boolean(d['clip_after_nms']) if 'clip_after_nms' in d else False
    clip_before = common_def.string_to_boolean(d['clip_before_nms']) if 'clip_before_nms' in d else False
    normalized = common_def.string_to_boolean(d['normalized']) if 'normalized' in d else False

    loc, conf, priors = (dev.as_device(inputs[p]) for p in (0, 1, 2))
    assert priors.shape[1] == 2                       # boxes and variances (DetectionOutput.py:177)
    assert share_location and normalized              # num_loc_classes == 1 (:189), normalized boxes (:220)
    if code_type not in ('caffe.PriorBoxParameter.CORNER', 'caffe.PriorBoxParameter.CENTER_SIZE'):
        raise ValueError('unknown code_type {!r}'.format(code_type))
    n = loc.shape[0]
    num_priors = priors.shape[2] // 4
    assert loc.shape[1] == num_priors * 4 and conf.shape[1] == num_priors * num_classes
    if keep_top_k[0] > 0:                             # output shape rule, :225-231
        records = keep_top_k[0]
    elif keep_top_k[0] == -1 and top_k > 0:
        records = top_k * num_classes
    else:
        records = num_classes * num_priors
    out = dev.DeviceTensor.empty((1, 1, n * records, 7))
    dev.call('pvhip_detection_output_f32', ctypes.c_void_p(loc.ptr), ctypes.c_void_p(conf.ptr), ctypes.c_void_p(priors.ptr),
             ctypes.c_void_p(out.ptr), n, num_priors, num_classes, records, confidence_threshold, nms_threshold,
             1 if code_type.endswith('CENTER_SIZE') else 0, int(variance_encoded), int(clip_before), int(clip_after))
    return {common_def.first_output_port(node): out}
