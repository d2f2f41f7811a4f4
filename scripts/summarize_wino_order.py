#!/usr/bin/env python3
"""Local: gpurun_out/wino_order/{2,1} (scripts/traffic_wino_order.sh) -> per layer the HBM bytes read by conv_wino4s_kernel in the two tile orders
(FETCH_SIZE in KB, doubled as MI355X_MICROARCH.md prescribes for gfx950) beside the algorithmic read bytes (input + transformed weights)."""
import csv, glob, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAYERS = [('4a/3x3', (256, 96, 14, 14), 208), ('4b/3x3', (256, 112, 14, 14), 224), ('4c/3x3', (256, 128, 14, 14), 256), ('4d/3x3', (256, 144, 14, 14), 288),
          ('4e/3x3', (256, 160, 14, 14), 320), ('5a/3x3', (256, 160, 7, 7), 320), ('5b/3x3', (256, 192, 7, 7), 384)]
res = {}
for o in ('2', '1'):
    f = glob.glob(os.path.join(REPO, 'gpurun_out', 'wino_order', o, '**', '*counter_collection.csv'), recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if r['Counter_Name'] == 'FETCH_SIZE' and 'conv_wino4s_kernel' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    vals = [float(r['Counter_Value']) * 1024 * 2 for r in rows]
    res[o] = [sum(vals[3 * i + 1:3 * i + 3]) / 2 for i in range(len(LAYERS))]          # the second and third launch of a layer
print('| layer | algorithmic read MB (input + U) | block-major MB | pair-major MB |')
print('|---|---|---|---|')
for i, (name, (n, c, h, w), k) in enumerate(LAYERS):
    alg = n * c * h * w * 4 + k * c * 36 * 4
    print('| {} | {:.1f} + {:.1f} | {:.1f} | {:.1f} |'.format(name, n * c * h * w * 4 / 1e6, k * c * 36 * 4 / 1e6, res['2'][i] / 1e6, res['1'][i] / 1e6))
