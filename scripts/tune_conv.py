#!/usr/bin/env python3
"""GPU-box tool: time every convolution tile configuration on every distinct Convolution shape of a model
(default googlenet-v1, batch 256) through the C ABI, and print / save the table.  Used to calibrate the tile
selection heuristic in pvhip_conv2d_f32; not part of the product path.

    python scripts/tune_conv.py [--model googlenet-v1] [--batch 256] [--reps 5] [--tiles 32x128,...]
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyopenvino_amd import IECore, device as dev, synth  # noqa: E402
from pyopenvino_amd import common_def  # noqa: E402

ALL_TILES = ['32x128', '64x128', '64x256', 'w1x1', 'w1x2', 'w2x1', 'w2x2', 'w4x1', 'w1x4']   # wAxB = wave-direct kernel, tile in units of 32


def conv_shapes(model, batch):
    ie = IECore()
    xml = os.path.join(REPO, 'models', model + '.xml')
    net = ie.read_network(xml, weights=bytes(64 << 20))
    net.set_batch(batch)
    seen, out = set(), []
    for nid in net.G.nodes:
        node = net.G.nodes[nid]
        if node['type'] != 'Convolution':
            continue
        a = node['data']
        key = (node['input'][0]['dims'], node['input'][1]['dims'], a['strides'], a['pads_begin'], a['pads_end'])
        if key in seen:
            continue
        seen.add(key)
        out.append((nid, node['name'], key))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='googlenet-v1')
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--tiles', default=','.join(ALL_TILES))
    ap.add_argument('--only', default='', help='substring filter on layer names')
    args = ap.parse_args()
    tiles = args.tiles.split(',')
    dev.init(0)
    table = []
    for nid, name, (xs, ws, st, pb, pe) in conv_shapes(args.model, args.batch):
        if args.only and args.only not in name:
            continue
        n, c, h, w = xs
        k, _, kh, kw = ws
        sh, sw = common_def.string_to_tuple(st)
        pt, pl = common_def.string_to_tuple(pb)
        pbm, pr = common_def.string_to_tuple(pe)
        oh = (h + pt + pbm - kh) // sh + 1
        ow = (w + pl + pr - kw) // sw + 1
        x = dev.DeviceTensor.from_numpy(synth.normal(1, nid, n * c * h * w).astype(np.float32).reshape(xs))
        wt = dev.DeviceTensor.from_numpy((synth.normal(2, nid, k * c * kh * kw) * 0.05).astype(np.float32).reshape(ws))
        y = dev.DeviceTensor.empty((n, k, oh, ow))
        elems = dev.call('pvhip_conv2d_pack_elems', k, c, kh, kw)
        wp = dev.DeviceTensor.empty((int(elems),))
        gflop = 2.0 * n * k * oh * ow * c * kh * kw / 1e9
        row = {'id': nid, 'name': name, 'x': xs, 'w': ws, 'gflop': gflop, 'ms': {}}

        def run():
            dev.call('pvhip_conv2d_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(wp.ptr), ctypes.c_void_p(y.ptr),
                     n, c, h, w, k, kh, kw, oh, ow, sh, sw, pt, pl, ctypes.c_void_p(0), 0, 0, 0, 0.0, 0.0)

        for tile in tiles + ['auto']:
            for envk in ('PVHIP_CONV_TILE', 'PVHIP_CONV_KERNEL', 'PVHIP_CONV_WTILE', 'PVHIP_CONV_ABLATE'):
                os.environ.pop(envk, None)
            os.environ['PVHIP_CONV_WINOGRAD'] = '1' if tile == 'auto' or tile.startswith('wg') else '0'     # other names are the direct kernels
            os.environ.pop('PVHIP_WINO_KB', None)
            os.environ.pop('PVHIP_WINO_WAVES', None)
            os.environ.pop('PVHIP_WINO_SMALL', None)
            os.environ.pop('PVHIP_CONV_NOPW', None)
            os.environ.pop('PVHIP_CONV_LDS_PAD_KB', None)
            if tile == 'auto':           # the library's own kernel / tile choice
                pass
            elif tile.startswith('wg'):      # wg<32|64>[x4]: Winograd, output channels per workgroup, 4 instead of 8 waves
                os.environ['PVHIP_WINO_KB'] = tile[2:4]
                if 'x4' in tile:
                    os.environ['PVHIP_WINO_WAVES'] = '4'
                os.environ['PVHIP_WINO_SMALL'] = '1' if tile.endswith('s') else '0'     # wg32s: 32 channels x 32 patches, four waves
            elif tile.startswith('p'):     # p<KB>:<tile>: LDS kernel with extra dynamic LDS (occupancy cap)
                kb, tl = tile[1:].split(':')
                os.environ['PVHIP_CONV_LDS_PAD_KB'] = kb
                os.environ['PVHIP_CONV_TILE'] = tl.lstrip('d')
                if not tl.startswith('d'):
                    os.environ['PVHIP_CONV_KERNEL'] = 'lds'
            elif tile.startswith('a'):     # a<bits>w<tile>: ablated wave kernel (diagnostic)
                os.environ['PVHIP_CONV_KERNEL'] = 'wave'
                os.environ['PVHIP_CONV_ABLATE'] = tile[1]
                os.environ['PVHIP_CONV_WTILE'] = tile[3:]
            elif tile.startswith('n'):     # n<tile>: LDS-DMA kernel without the 16-byte pointwise copy
                os.environ['PVHIP_CONV_KERNEL'] = 'dma'
                os.environ['PVHIP_CONV_TILE'] = tile[1:]
                os.environ['PVHIP_CONV_NOPW'] = '1'
            elif tile.startswith("d"):     # d<tile>: LDS-DMA kernel (the default for (r,s)-major shapes)
                os.environ['PVHIP_CONV_KERNEL'] = 'dma'
                os.environ['PVHIP_CONV_TILE'] = tile[1:]
            elif tile.startswith('w'):
                os.environ['PVHIP_CONV_KERNEL'] = 'wave'
                os.environ['PVHIP_CONV_WTILE'] = tile[1:]
            elif tile != 'auto':
                os.environ['PVHIP_CONV_KERNEL'] = 'lds'
                os.environ['PVHIP_CONV_TILE'] = tile
            dev.call('pvhip_conv2d_pack_f32', ctypes.c_void_p(wt.ptr), ctypes.c_void_p(wp.ptr), k, c, kh, kw, h, w)   # layout follows the env
            run()
            dev.synchronize()
            e0 = dev.Event().record()
            for _ in range(args.reps):
                run()
            e1 = dev.Event().record()
            e1.synchronize()
            row['ms'][tile] = e0.elapsed_ms(e1) / args.reps
        for envk in ('PVHIP_CONV_TILE', 'PVHIP_CONV_KERNEL', 'PVHIP_CONV_WTILE', 'PVHIP_CONV_WINOGRAD', 'PVHIP_WINO_KB', 'PVHIP_WINO_WAVES'):
            os.environ.pop(envk, None)
        best = min(tiles, key=lambda t: row['ms'][t])
        row['best'] = best
        table.append(row)
        print('{:4d} {:36s} x{} w{} gflop {:7.2f} | '.format(nid, name[-36:], xs, ws, gflop) +
              ' '.join('{}:{:.3f}'.format(t, row['ms'][t]) for t in tiles) +
              ' | auto {:.3f} best {} {:.3f} ({:.1f} TF)'.format(row['ms']['auto'], best, row['ms'][best], gflop / row['ms'][best]),
              flush=True)
    tot_auto = sum(r['ms']['auto'] for r in table)
    tot_best = sum(r['ms'][r['best']] for r in table)
    print('sum over distinct shapes: auto {:.3f} ms, best-per-layer {:.3f} ms'.format(tot_auto, tot_best))
    out_dir = os.path.join(REPO, 'gpurun_out')
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, 'tune_conv.json'), 'w') as f:
            json.dump(table, f, indent=1)


if __name__ == '__main__':
    main()
