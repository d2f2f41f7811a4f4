# Host-side helper shared by Add / Multiply: resolve numpy broadcasting into element strides and
# launch the binary kernel.  (Leading underscore: not a plugin, skipped by the loader.)
import ctypes

from .. import device as dev


def strides_for_broadcast(shape, target_shape):
    """Element strides of a contiguous tensor of `shape` viewed at `target_shape` under
    np.broadcast_to rules (right-aligned, extent-1 axes repeat).  Raises ValueError like numpy."""
    shape, target_shape = tuple(shape), tuple(target_shape)
    if len(shape) > len(target_shape):
        raise ValueError('input operand has more dimensions than allowed by the axis remapping')
    contiguous, acc = [], 1
    for d in reversed(shape):
        contiguous.append(acc)
        acc *= d
    contiguous.reverse()
    lead = len(target_shape) - len(shape)
    out = [0] * len(target_shape)
    for i, (d, st) in enumerate(zip(shape, contiguous)):
        t = target_shape[lead + i]
        if d == t:
            out[lead + i] = st if d != 1 else 0
        elif d == 1:
            out[lead + i] = 0
        else:
            raise ValueError('operands could not be broadcast together with remapped shapes '
                             '[original->remapped]: {} and requested shape {}'.format(shape, target_shape))
    return out


def launch(entry: str, a, b, out_shape):
    """out = a (op) b, both broadcast to out_shape."""
    out_shape = tuple(int(d) for d in out_shape)
    if len(out_shape) > dev.MAX_RANK:
        raise NotImplementedError('rank {} > {}'.format(len(out_shape), dev.MAX_RANK))
    a_st = strides_for_broadcast(a.shape, out_shape)
    b_st = strides_for_broadcast(b.shape, out_shape)
    out = dev.DeviceTensor.empty(out_shape)
    dev.call(entry, ctypes.c_void_p(a.ptr), ctypes.c_void_p(b.ptr), ctypes.c_void_p(out.ptr), len(out_shape),
             dev.i64_array(out_shape), dev.i64_array(a_st), dev.i64_array(b_st))
    return out
