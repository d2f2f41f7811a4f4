# ORACLE -- test infrastructure only.  PriorBoxClustered: CPU restatement of reference
# op_plugins/PriorBoxClustered.py:43-68 (attribute parsing incl. its int() on the step attributes).
from .. import ops
from ._util import check, out_port


def name():
    print('PriorBoxClustered')


def floats(text):
    return [float(t) for t in text.split(',')]


def compute(node: dict, inputs: dict = None, kernel_type: str = 'special', debug: bool = False):
    check(node, inputs)
    d = node['data']
    res = ops.prior_box_clustered(
        inputs[0], inputs[1],
        width=floats(d['width']) if 'width' in d else [1.0], height=floats(d['height']) if 'height' in d else [1.0],
        step=int(d['step']) if 'step' in d else 0.0, step_h=int(d['step_h']) if 'step_h' in d else 0.0,
        step_w=int(d['step_w']) if 'step_w' in d else 0.0, offset=float(d['offset']),
        variance=floats(d['variance']) if 'variance' in d else [],
        img_h=float(d['img_h']) if 'img_h' in d else 0.0, img_w=float(d['img_w']) if 'img_w' in d else 0.0)
    return {out_port(node): res}
