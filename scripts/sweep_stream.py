#!/usr/bin/env python3
"""GPU-box tool (diagnostic build: `make -C pyopenvino_amd/csrc diag`): what can a float4 copy / ReLU stream reach on THIS box?

Sweeps, through libpvhip_diag.so's pvhip_diag_stream_f32, the shape of a streaming kernel -- 16-byte loads per lane issued before the
first store (1/2/4/8), where those loads lie (a grid apart, adjacent pieces of a workgroup, adjacent 16-byte words of a lane), plain or
nontemporal loads and stores, workgroups per CU (1..16 x 256 threads; 512- and 1024-thread workgroups), tensor size -- against
hipMemcpyDtoD and the product's ReLU (pvhip_relu_f32) in the same process.  Rates are (bytes read + bytes written) / time, the
convention of `roofline` for the memory-bound ops; the hardware guide's float4 copy is 6.29 TB/s (MI355X_MICROARCH.md:36).

    python scripts/sweep_stream.py [--out gpurun_out/stream_sweep.md] [--reps 10]
"""
import argparse, ctypes, itertools, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(REPO, 'gpurun_out', 'stream_sweep.md'))
    ap.add_argument('--reps', type=int, default=10)
    args = ap.parse_args()
    dev.LIB_PATH = os.path.join(os.path.dirname(dev.LIB_PATH), 'libpvhip_diag.so')
    dev.init(0)
    lib = ctypes.CDLL(dev.LIB_PATH)
    fn = lib.pvhip_diag_stream_f32
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulonglong] + [ctypes.c_int] * 6
    fn.restype = ctypes.c_int

    def timeit(run, reps=args.reps):
        run(); run(); dev.synchronize()
        best = 1e9
        for _ in range(3):
            e0 = dev.Event().record()
            for _ in range(reps):
                run()
            e1 = dev.Event().record(); e1.synchronize()
            best = min(best, e0.elapsed_ms(e1) / reps)
        return best

    lines = ['# float4 stream sweep (`scripts/sweep_stream.py`, diagnostic build), device: {}'.format(dev.device_name()), '',
             'Rates in TB/s of bytes read + written.  `U` = 16-byte loads per lane issued before the first store; layout `grid` = the U loads '
             'of a lane are a whole grid apart, `wg` = U adjacent pieces of one workgroup (a contiguous run of U x 4 KiB at 256 threads), '
             '`lane` = U adjacent 16-byte words per lane; `nt` = nontemporal loads and stores; `wg/CU` = workgroups per CU (x threads).', '']
    sizes = [(256, 64, 56, 56), (256, 64, 112, 112), (256, 192, 112, 112)]      # 0.41, 1.64, 4.9 GB moved
    bufs = {}
    summary = []
    for shape in sizes:
        n = int(np.prod(shape))
        x = dev.DeviceTensor.empty(shape); y = dev.DeviceTensor.empty(shape)
        dev.call('pvhip_memset', ctypes.c_void_p(x.ptr), 0x3f, n * 4)
        gb = 8.0 * n / 1e9
        t_cp = timeit(lambda: dev.call('pvhip_memcpy_d2d', ctypes.c_void_p(y.ptr), ctypes.c_void_p(x.ptr), n * 4))
        t_relu = timeit(lambda: dev.call('pvhip_relu_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(y.ptr), n))
        lines += ['## tensor {} = {:.2f} GB moved: hipMemcpyDtoD {:.2f} TB/s, product ReLU (pvhip_relu_f32) {:.2f} TB/s'.format(
            shape, gb, gb / t_cp, gb / t_relu), '', '| mode | U | layout | nt | threads | wg/CU | ms | TB/s |', '|---|---|---|---|---|---|---|---|']
        rows = []
        for relu, U, layout, nt, (threads, wpc) in itertools.product(
                (0, 1), (1, 2, 4, 8), (0, 1, 2), (0, 1),
                ((256, 1), (256, 2), (256, 4), (256, 8), (256, 16), (256, 32), (512, 4), (1024, 2))):
            if layout == 2 and U == 1:
                continue
            if relu and (threads != 256 or wpc not in (4, 8, 16)):
                continue
            blocks = 256 * wpc

            def run():
                rc = fn(x.ptr, y.ptr, n, relu, U, nt, layout, blocks, threads)
                assert rc == 0, dev.call('pvhip_last_error') if False else rc
            t = timeit(run)
            rows.append((gb / t, relu, U, layout, nt, threads, wpc, t))
        rows.sort(reverse=True)
        for rate, relu, U, layout, nt, threads, wpc, t in rows[:24] + rows[-4:]:
            lines.append('| {} | {} | {} | {} | {} | {} | {:.3f} | {:.2f} |'.format('relu' if relu else 'copy', U, ('grid', 'wg', 'lane')[layout], nt, threads, wpc, t, rate))
        best_copy = max(r for r in rows if not r[1]); best_relu = max(r for r in rows if r[1])
        # the product shape (U=1, grid, plain, 256 threads, 8 per CU) for reference
        prod = [r for r in rows if r[1:7] == (1, 1, 0, 0, 256, 8)]
        summary.append('| {} | {:.2f} | {:.2f} | {:.2f} | {:.2f} (U={} {} nt={} {}x{}/CU) | {:.2f} (U={} {} nt={} {}x{}/CU) | {:.2f} |'.format(
            shape, gb, gb / t_cp, gb / t_relu, best_copy[0], best_copy[2], ('grid', 'wg', 'lane')[best_copy[3]], best_copy[4], best_copy[5], best_copy[6],
            best_relu[0], best_relu[2], ('grid', 'wg', 'lane')[best_relu[3]], best_relu[4], best_relu[5], best_relu[6], prod[0][0] if prod else float('nan')))
        lines.append('')
        print('\n'.join(lines[-34:]), flush=True)
        del x, y
    head = ['## summary', '', '| tensor | GB moved | hipMemcpyDtoD | product ReLU | best copy | best ReLU | ReLU in the product shape (U=1 grid 256x8/CU) |',
            '|---|---|---|---|---|---|---|'] + summary + ['']
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, 'w') as f:
        f.write('\n'.join(lines[:4] + head + lines[4:]) + '\n')
    print('\n'.join(head))


if __name__ == '__main__':
    main()
