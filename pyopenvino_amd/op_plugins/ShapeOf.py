# ShapeOf -- HIP plugin.  Replaces reference op_plugins/ShapeOf.py:10-25: the input's dims as an integer vector.
# Shape arithmetic stays on the host (the values never depend on tensor data); nothing is launched.
import numpy as np

from .. import common_def


# A pass made of such nodes can be recorded into a hipGraph on ONE stream (Executable_Network.infer does so by itself for
# device-resident inputs; the whole SSD IR was tried: scripts/repro_capture.py).
GRAPH_CAPTURE_SAFE = True

def name():
    print('ShapeOf')


def compute(node: dict, inputs: dict = None, kernel_type: str = 'hip', debug: bool = False):
    if debug:
        print(node)
    common_def.validate_inputs(node, inputs)
    port = common_def.first_output_port(node)
    dims = node['input'][next(iter(node['input']))]['dims']
    return {port: np.array(dims, dtype=common_def.type_convert_tbl[node['output'][port]['precision']])}
