#!/usr/bin/env python3
"""GPU-box tool: the pointwise (1x1) convolutions of googlenet-v1 at batch 256 -- the nine sibling launches (1x1 + 3x3_reduce +
5x5_reduce of a module as one launch), the pool_proj layers and conv2/3x3_reduce -- on the pointwise kernel (pvhip_pw.hip) and on
the general LDS-DMA kernel (PVHIP_CONV_POINTWISE=0), through the C ABI: time per launch, TFLOP/s, and whether the two agree bit
for bit.  Interleaved rounds in one process (same box, same clocks).

    python scripts/time_pw.py [--reps 20] [--rounds 3] [--tiles 0,50,100]
"""
import argparse, ctypes, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import device as dev, synth  # noqa: E402

# (name, C, HW side, [K of each member])
SHAPES = [
    ('conv2/3x3_reduce', 64, 56, [64]),
    ('3a siblings', 192, 28, [64, 96, 16]), ('3b siblings', 256, 28, [128, 128, 32]),
    ('4a siblings', 480, 14, [192, 96, 16]), ('4b siblings', 512, 14, [160, 112, 24]), ('4c siblings', 512, 14, [128, 128, 24]),
    ('4d siblings', 512, 14, [112, 144, 32]), ('4e siblings', 528, 14, [256, 160, 32]),
    ('5a siblings', 832, 7, [256, 160, 32]), ('5b siblings', 832, 7, [384, 192, 48]),
    ('3a pool_proj', 192, 28, [32]), ('3b pool_proj', 256, 28, [64]), ('4a pool_proj', 480, 14, [64]), ('4e pool_proj', 528, 14, [128]),
    ('5a pool_proj', 832, 7, [128]),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--tiles', default='100', help='PVHIP_PW_STAGGER values to try (percent of the library rule, 0 = no stagger)')
    ap.add_argument('--only', default='')
    ap.add_argument('--ablate', default='', help='PVHIP_PW_ABLATE values to time (needs `make -C pyopenvino_amd/csrc diag`: libpvhip_diag.so is loaded then)')
    args = ap.parse_args()
    if args.ablate:
        dev.LIB_PATH = os.path.join(os.path.dirname(dev.LIB_PATH), 'libpvhip_diag.so')
    dev.init(0)
    n = args.batch
    variants = [('general', {'PVHIP_CONV_POINTWISE': '0'})] + [('pw tn ' + t, {'PVHIP_CONV_POINTWISE': '1', 'PVHIP_PW_TN': t}) for t in args.tiles.split(',')]
    variants = variants + [('pw, three stage buffers', {'PVHIP_CONV_POINTWISE': '1', 'PVHIP_TUNE2': '3'})]
    variants += [('pw128 ablate ' + v, {'PVHIP_CONV_POINTWISE': '1', 'PVHIP_PW_STAGGER': '0', 'PVHIP_PW_ABLATE': v}) for v in args.ablate.split(',') if v]
    total = {v[0]: 0.0 for v in variants}
    for name, c, side, ks in SHAPES:
        if args.only and args.only not in name:
            continue
        hw = side * side
        x = dev.DeviceTensor.from_numpy(synth.normal(1, c, n * c * hw).astype(np.float32).reshape(n, c, side, side))
        pads = [-(-k // 32) * 32 for k in ks]
        wh = np.zeros((sum(pads), c, 1, 1), np.float32)
        bh = np.zeros((sum(pads),), np.float32)
        row = 0
        for k, kp in zip(ks, pads):
            wh[row:row + k] = (synth.normal(2, c + k, k * c) * 0.05).astype(np.float32).reshape(k, c, 1, 1)
            bh[row:row + k] = synth.normal(3, k, k).astype(np.float32)
            row += kp
        wf, bias = dev.DeviceTensor.from_numpy(wh), dev.DeviceTensor.from_numpy(bh)
        elems = dev.call('pvhip_conv2d_pack_elems', wh.shape[0], c, 1, 1)
        wp = dev.DeviceTensor.empty((int(elems),))
        dev.call('pvhip_conv2d_pack_f32', ctypes.c_void_p(wf.ptr), ctypes.c_void_p(wp.ptr), wh.shape[0], c, 1, 1, side, side)
        outs = [dev.DeviceTensor.empty((n, k, side, side)) for k in ks]
        dests = (dev.ConvDest * len(ks))()
        for i, (k, o) in enumerate(zip(ks, outs)):
            dests[i].y, dests[i].k, dests[i].channel_offset, dests[i].channels_total = o.ptr, k, 0, 0

        def run():
            dev.call('pvhip_conv2d_multi_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(wp.ptr), n, c, side, side, 1, 1, side, side, 1, 1, 0, 0,
                     ctypes.c_void_p(bias.ptr), 1, 0.0, 0.0, len(ks), ctypes.cast(dests, ctypes.c_void_p))

        gflop = 2.0 * n * hw * c * sum(ks) / 1e9
        best = {v[0]: 1e9 for v in variants}
        ref = None
        same = {}
        stamp_note = ''
        for rnd in range(args.rounds):
            for vname, env in variants:
                for kenv in ('PVHIP_CONV_POINTWISE', 'PVHIP_PW_STAGGER', 'PVHIP_PW_TN', 'PVHIP_PW_ABLATE', 'PVHIP_TUNE2'):
                    os.environ.pop(kenv, None)
                os.environ.update(env)
                dev.reload_settings()
                run(); dev.synchronize()
                if env.get('PVHIP_PW_ABLATE') == '8':
                    lib0 = ctypes.CDLL(dev.LIB_PATH); lib0.pvhip_diag_pw_stamps.argtypes = [ctypes.c_void_p]; lib0.pvhip_diag_pw_stamps((ctypes.c_ulonglong * 8)())
                if rnd == 0:
                    got = [o.numpy() for o in outs]
                    if ref is None:
                        ref = got
                    same[vname] = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(got, ref))
                    for o in outs:
                        dev.call('pvhip_memset', ctypes.c_void_p(o.ptr), 0xff, o.nbytes)
                e0 = dev.Event().record()
                for _ in range(args.reps):
                    run()
                e1 = dev.Event().record(); e1.synchronize()
                best[vname] = min(best[vname], e0.elapsed_ms(e1) / args.reps)
                if env.get('PVHIP_PW_ABLATE') == '8' and rnd == args.rounds - 1:      # s_memtime stamps of every 61st workgroup's wave 0
                    lib = ctypes.CDLL(dev.LIB_PATH)
                    lib.pvhip_diag_pw_stamps.argtypes = [ctypes.c_void_p]
                    st = (ctypes.c_ulonglong * 8)()
                    lib.pvhip_diag_pw_stamps(st)
                    cnt = max(1, st[3])
                    stamp_note = '  [per workgroup: prologue {:.0f}, main loop {:.0f}, epilogue {:.0f} cycles (of which {:.0f} until the bias has arrived); {:.2f} GHz]'.format(
                        st[0] / cnt, st[1] / cnt, st[2] / cnt, st[5] / cnt, (st[0] + st[1] + st[2]) / max(1, st[4]) / 10.0)
        line = '{:18s} C={:4d} {:2d}x{:<2d} K={:<14s} {:6.2f} GF |'.format(name, c, side, side, '+'.join(map(str, ks)), gflop)
        for vname, _ in variants:
            total[vname] += best[vname]
            line += ' {}: {:.4f} ms {:5.1f} TF {} |'.format(vname, best[vname], gflop / best[vname], 'same' if same[vname] else 'DIFF')
        print(line + stamp_note, flush=True)
    print('total ms: ' + '  '.join('{} {:.3f}'.format(k, v) for k, v in total.items()), flush=True)


if __name__ == '__main__':
    main()
