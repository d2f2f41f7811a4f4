# ORACLE -- test infrastructure only.  DetectionOutput: CPU restatement of reference
# op_plugins/DetectionOutput.py:272-305 (attributes) and :163-259 (kernel).  Batch rule of this build (the
# reference asserts N == 1): images are processed independently and image b's records are rows
# [b * records, (b + 1) * records) of the output; column 0 is the record index inside the image, as in the reference.
import numpy as np

from .. import ops
from ._util import check, ints, out_port


def name():
    print('DetectionOutput')


def truth(text):
    return text.strip().lower() in ('true', '1')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'special', debug: bool = False):
    check(node, inputs)
    d = node['data']
    loc, conf, priors = inputs[0], inputs[1], inputs[2]
    assert priors.shape[1] == 2
    assert truth(d.get('share_location', 'true')) and truth(d.get('normalized', 'false'))
    per_image = [ops.detection_output(
        loc[b], conf[b], priors, num_classes=int(d['num_classes']), keep_top_k=ints(d['keep_top_k'])[0],
        top_k=int(d.get('top_k', -1)), code_type=d.get('code_type', 'caffe.PriorBoxParameter.CORNER'),
        variance_encoded=truth(d.get('variance_encoded_in_target', 'false')), nms_threshold=float(d['nms_threshold']),
        confidence_threshold=float(d.get('confidence_threshold', 0)), clip_before_nms=truth(d.get('clip_before_nms', 'false')),
        clip_after_nms=truth(d.get('clip_after_nms', 'false'))) for b in range(loc.shape[0])]
    return {out_port(node): np.concatenate(per_image, 0)[None, None]}
