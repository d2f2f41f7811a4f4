#!/usr/bin/env python3
"""Benchmark of the hot path: images/sec of googlenet-v1 fp32 at batch 256 per GPU (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole per-layer compute() path (Executable_Network.infer) over one batch of
256 synthetic images per GPU, input already resident in HBM, ending with the Result tensor back on the
host (and, for N > 1, an RCCL all-gather of the Result tensors first).  For N > 1 there is one process per GPU:
either the driver starts them (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`: RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or `python bench.py --gpus N` starts them itself
(shard.launch_ranks: N fresh child processes with the same variables; the parent never touches a GPU).  Rendezvous,
barrier and max-over-ranks go over plain sockets (shard.TcpGroup); no torch in the process.

Parity of the timed path: rows 0-7 of every rank's request 0 are the eight images of tests/golden/googlenet_rows8.npz (the
reference's own N=1 answers); after the timed region the LAST replayed Result of every request is checked -- request 0's
rows 0-7 against the fixture (1e-4 in both norms), every request bit for bit against one eager synchronous infer() of the
same tensor -- and the line says so (`parity_of_timed_path`).  A mismatch fails the run.

The timed region is a block of exactly K steps between barrier + device synchronisation on both sides (max over
ranks); the block is repeated until at least --min-seconds (1 s) have been timed and the MEDIAN block is reported
(`steps` = K, `ms_per_step` = median block / K, `timed_blocks`, `timed_region_s`).  Every third block is instrumented (one of its
steps runs alone with hipEvent brackets: the roofline measurement, inside the timed region); `ms_per_step_instrumented_blocks`
reports those blocks' own median.

Rank 0 prints ONE JSON line (contract in the task statement) carrying
  roofline      for the dominant kernels, the fp32-MFMA Convolution launches (Winograd forms, pointwise, implicit GEMM):
                `achieved` / `frac` = the flops the matrix cores EXECUTE per launch (Winograd families: 16/36, 36/144, 36/100 of the
                algorithmic count, padded patches included) / the launches' device time, measured with hipEvents on the compute
                stream inside the timed blocks, against the 157.3 TFLOP/s fp32 MFMA peak; `achieved_algorithmic` /
                `frac_algorithmic` = SURVEY 8(d)'s algorithmic flops (2*N*K*C*kh*kw*oh*ow) over the same time (how fast the layers
                get done; may exceed 1); `per_kernel` = per kernel family with its own bound min(MFMA peak, arithmetic intensity x
                HBM peak); `traffic` / `traffic_by_kernel` = HBM bytes per launch from the committed PMC passes;
  cpu_baseline  the oracle (CPU restatement of the reference's 'special' path) timed on this host, N=1 per
                image like the reference, on a bounded sample of the same workload;
  single_request_images_per_sec   one synchronous infer() at a time (B / wall time of one infer, median);
  roofline.sustained              what fp32 MFMA alone sustains on THIS box (TFLOP/s, shader clock) and executed flops against it;
  hbm_copy_ceiling_GBs            what a plain device-to-device copy and the 16-byte ReLU stream reach on THIS box;
  extra_configs                   BASELINE configs 2 (mnist batch 64) and 5 (ssd_mobilenet_v1_coco batch 128).
A per-op-type breakdown goes to stderr and, if the directory exists, the per-layer table to gpurun_out/bench_layers.json
(committed per round as profiles/rNN_layers.json).
"""
import argparse
import ctypes
import glob
import json
import os
import statistics
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

MODEL = 'googlenet-v1'
BATCH_PER_GPU = 256
WEIGHT_SEED = 1234
PEAK_MFMA_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 MFMA dense peak (2.4 GHz)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak
SAMPLE_EVERY = 20              # every 20th step of a timed block (the first included) runs alone, with event brackets
KERNEL_NODES = {'Convolution', 'MatMul', 'MaxPool', 'AvgPool', 'Add', 'Multiply', 'ReLU', 'SoftMax', 'LRN',
                'Concat', 'Transpose', 'GroupConvolution', 'Clamp', 'Sigmoid', 'DetectionOutput'}


def node_work(node, inputs_shapes, out_shape):
    """Algorithmic work of one node: (flops, bytes) per SURVEY section 8(d): conv/matmul flops =
    2*MACs; memory-bound ops bytes = 4*(elements read once + written once), broadcast operand once."""
    t = node['type']
    out_e = int(np.prod(out_shape)) if len(out_shape) else 1
    in_e = [int(np.prod(s)) if len(s) else 1 for s in inputs_shapes]
    if t == 'Convolution':
        k, c, kh, kw = inputs_shapes[1]
        return 2.0 * out_e * c * kh * kw, 4.0 * (in_e[0] + in_e[1] + out_e)
    if t == 'MatMul':
        kdim = inputs_shapes[0][-1]
        return 2.0 * out_e * kdim, 4.0 * (sum(in_e) + out_e)
    return 0.0, 4.0 * (sum(in_e) + out_e)


def collect_work(net):
    G = net.G
    work = {}
    for nid in G.nodes:
        node = G.nodes[nid]
        if node['type'] in ('Const', 'Parameter', 'Result') or 'output' not in node:
            continue
        ins = [node['input'][p]['dims'] for p in sorted(node.get('input', {}))]
        if node['type'] in ('LRN', 'Reshape', 'Transpose'):
            ins = ins[:1]
        out = next(iter(node['output'].values()))['dims']
        work[nid] = node_work(node, ins, out)
    return work


def rows_vs_reference(gathered, golden_out, batch, world, n_gold, parity):
    """The golden rows of EVERY rank in the gathered Result of request 0: rank w's shard starts at row w * batch and its first n_gold rows are the
    images the reference answered at N=1 (every rank feeds the same golden images: the weights are replicated, so every rank must return the
    same answers).  Accumulates into `parity` (max-norm error and the worst element's share of the 1e-4 allowance); a two-rank CPU test
    drives exactly this function with a gathered tensor (tests/test_shard_gloo.py)."""
    gathered = np.asarray(gathered)
    assert gathered.shape[0] == batch * world, (gathered.shape, batch, world)
    want = np.asarray(golden_out[:n_gold], dtype=np.float64)
    for w in range(world):
        rows = gathered[w * batch: w * batch + n_gold].astype(np.float64)
        err = float(np.abs(rows - want).max() / np.abs(want).max())
        rms = float(np.sqrt(np.mean(want * want)))
        excess = float((np.abs(rows - want) / (1e-4 * np.abs(want) + 1e-4 * rms)).max())
        parity['rows_vs_reference'] += n_gold
        parity['max_norm_error_vs_reference'] = max(parity['max_norm_error_vs_reference'] or 0.0, err)
        parity['worst_element_of_1e-4_allowance'] = max(parity['worst_element_of_1e-4_allowance'] or 0.0, excess)
    return parity


def cpu_baseline(blob, n_images):
    """Oracle plugins, one image at a time (the only mode the reference supports)."""
    from pyopenvino_amd import IECore, synth
    ie = IECore(plugin_package='oracle.op_plugins')
    net = ie.read_network(os.path.join(REPO, 'models', MODEL + '.xml'), weights=blob)
    ex = ie.load_network(net)
    ex.kernel_type = 'special'
    name = net.inputs[0]['name']
    ex.infer({name: synth.uniform_pixels(1, (1, 3, 224, 224))})   # warm-up (page-in, BLAS threads)
    t0 = time.time()
    for i in range(n_images):
        ex.infer({name: synth.uniform_pixels(2 + i, (1, 3, 224, 224))})
    dt = time.time() - t0
    threads = os.cpu_count()
    try:
        from threadpoolctl import threadpool_info
        blas = [p for p in threadpool_info() if p.get('user_api') == 'blas']
        if blas:
            threads = int(blas[0]['num_threads'])
    except Exception:
        pass
    return {'value': n_images / dt, 'unit': 'images/sec', 'cores': threads, 'kind': 'port',
            'sample': '{} googlenet-v1 images, one N=1 forward each, oracle numpy/OpenBLAS plugins, {:.1f} s'.format(n_images, dt)}


_diag = None


def diag_library(device):
    """The DIAGNOSTIC build (libpvhip_diag.so, include/pvhip_diag.h: measurement probes that are not part of the product) as a second
    library handle with its own state, initialised on the same device.  Harness only -- and only AFTER the timed region: while the
    benchmark is timed, libpvhip.so is the only library of this repository mapped into the process."""
    global _diag
    if _diag is None:
        path = device.DIAG_LIB_PATH
        if not os.path.isfile(path):
            raise RuntimeError('diagnostic build {} is missing -- run `make -C pyopenvino_amd/csrc diag`'.format(path))
        lib = ctypes.CDLL(path)
        lib.pvhip_last_error.restype = ctypes.c_char_p
        if lib.pvhip_init(int(device.current_device())) != 0:
            raise RuntimeError('diagnostic build: pvhip_init failed: {}'.format(lib.pvhip_last_error().decode(errors='replace')))
        lib.pvhip_mfma_ceiling_f32.restype = ctypes.c_int
        lib.pvhip_mfma_ceiling_f32.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        _diag = lib
    return _diag


def mfma_ceiling_f32(device, with_valu_partner=False, iters=20000):
    """(TFLOP/s, shader clock in GHz) this device sustains on v_mfma_f32_32x32x2_f32 alone -- or with a VALU-only wave beside every
    MFMA wave (include/pvhip_diag.h: pvhip_mfma_ceiling_f32)."""
    lib = diag_library(device)
    tf, ghz = ctypes.c_double(0.0), ctypes.c_double(0.0)
    rc = lib.pvhip_mfma_ceiling_f32(1 if with_valu_partner else 0, int(iters), ctypes.byref(tf), ctypes.byref(ghz))
    if rc != 0:
        raise RuntimeError('pvhip_mfma_ceiling_f32 failed ({}): {}'.format(rc, lib.pvhip_last_error().decode(errors='replace')))
    return float(tf.value), float(ghz.value)


def mfma_ceiling(device):
    """What THIS box sustains on v_mfma_f32_32x32x2_f32 with nothing else in the instruction stream (pvhip_mfma_ceiling_f32: one
    wave per SIMD, operands in registers, random data, a 2-3 ms kernel), the shader clock it holds meanwhile, and the same with a
    VALU-only wave beside every MFMA wave.  157.3 TFLOP/s is the rate at 2.4 GHz; the chip lowers its clock under matrix load."""
    # a probe of the DIAGNOSTIC build, loaded after the timed region for this measurement only
    try:
        diag_library(device)
    except Exception as exc:       # noqa: BLE001 -- informational: the line then says it has no such figure
        print('bench.py: roofline.sustained not measured: {}'.format(exc), file=sys.stderr)
        return None
    # the clock ramps up over the first milliseconds after idle: best of four back-to-back runs
    tf, ghz = max(mfma_ceiling_f32(device, False, 20000) for _ in range(4))
    tf_v, ghz_v = max(mfma_ceiling_f32(device, True, 20000) for _ in range(2))
    return {'TFLOPs': round(tf, 1), 'clock_GHz': round(ghz, 2), 'with_a_VALU_wave_per_SIMD_TFLOPs': round(tf_v, 1),
            'note': 'fp32 MFMA alone, every SIMD of every CU issuing; with an fp32 VALU wave on the same SIMD the MFMA rate drops by '
                    'that wave\'s share of the issue cycles: matrix and vector fp32 instructions of a SIMD do not overlap'}


def copy_ceiling(device):
    """Achieved HBM rate of a plain device-to-device copy and of the 16-byte-per-lane ReLU stream on this box, on a tensor
    far larger than the 256 MiB Infinity Cache (GoogLeNet's conv1 output, 822 MB): the ceiling the memory-bound ops of the
    pass can be held against (the spec's 8 TB/s is not reachable by any kernel)."""
    n = 256 * 64 * 112 * 112
    x, y = device.DeviceTensor.empty((n,)), device.DeviceTensor.empty((n,))
    device.call('pvhip_memset', ctypes.c_void_p(x.ptr), 0, n * 4)
    out = {}
    for name, fn in (('memcpy_d2d', lambda: device.call('pvhip_memcpy_d2d', ctypes.c_void_p(y.ptr), ctypes.c_void_p(x.ptr), n * 4)),
                     ('relu_stream', lambda: device.call('pvhip_relu_f32', ctypes.c_void_p(x.ptr), ctypes.c_void_p(y.ptr), n))):
        fn()
        device.synchronize()
        e0 = device.Event().record()
        for _ in range(10):
            fn()
        e1 = device.Event().record()
        e1.synchronize()
        out[name] = round(8.0 * n / (e0.elapsed_ms(e1) / 10 * 1e-3) / 1e9, 1)
    out['tensor_MB'] = round(4.0 * n / 1e6, 1)
    out['relu_stream_frac_hbm_peak'] = round(out['relu_stream'] / PEAK_HBM_GBS, 4)      # ReLU.py:9-12 on an 822 MB tensor against the 8 TB/s spec
    return out


def rocprof_per_op():
    """ms per step of the memory-bound launches from the committed rocprofv3 kernel statistics (profiles/r05_kernel_stats.csv: the one-stream bench command), keyed
    like per_op: average duration x launches per pass.  {} when the file is not there."""
    import csv
    path = os.path.join(REPO, 'profiles', 'r05_kernel_stats.csv')
    if not os.path.isfile(path):
        return {}
    families = {'MaxPool+LRN+1x1': ('maxpool3x3_lrn_conv1x1_kernel',), 'LRN+MaxPool': ('lrn_maxpool3x3_kernel',), 'MaxPool': ('maxpool3x3_cols_kernel', 'maxpool2d'),
                'AvgPool': ('avgpool2d_lds_kernel',), 'SoftMax': ('softmax_rows_kernel',), 'Add': ('binary_channel_kernel',), 'MatMul': ('matmul_kernel', 'matmul_reduce_kernel')}
    rows = list(csv.DictReader(open(path)))
    passes = max([int(r['Calls']) for r in rows if 'conv_stem_f32_kernel' in r['Name']] or [0])
    if passes == 0:
        return {}
    out = {}
    for typ, names in families.items():
        total = sum(float(r['TotalDurationNs']) for r in rows if any(nm in r['Name'] for nm in names))
        if total > 0:
            out[typ] = total / passes * 1e-6
    return out


def median_infer_rate(ex, feed, batch, reps, warm=3):
    from pyopenvino_amd import device
    for _ in range(warm):
        ex.infer(feed)
    times = []
    for _ in range(reps):
        device.synchronize()
        t0 = time.perf_counter()
        ex.infer(feed)
        times.append(time.perf_counter() - t0)
    return batch / statistics.median(times), statistics.median(times) * 1e3


PEAK_MFMA_F16_TFLOPS = 2500.0  # MI355X_MICROARCH.md: BF16 / FP16 MFMA dense peak (~2.5 PF; never the 2:1-sparsity figure)


def fp16_roofline(ex, net, feed):
    """The FP16 entry against ITS roofline: one per-layer pass (one hipEvent bracket per launch, one stream), the Convolution
    launches grouped by the kernel that ran (node['_hip_f16']) and by window; per family: launches, ms, algorithmic f16 flops (2*N*K*C*kh*kw*oh*ow:
    a direct contraction executes exactly those, plus the padding of its pixel and channel tiles, which is NOT counted: the fraction is useful work
    over the peak) against min(2.5 PFLOP/s, AI x 8 TB/s) with the bytes of fp16 tensors (2 per element in and out, 2 per weight).  `traffic` /
    `traffic_ratio`: HBM bytes per Convolution launch from the committed PMC passes of `python bench.py --fp16-passes N` (profiles/*_fp16_traffic.json)."""
    G = net.G
    work = collect_work(net)
    ex.device_timing, ex.compute_streams = 'all', 1
    ex.infer(feed)
    times = list(ex.device_times_ms())
    ex.device_timing = None
    fams, conv_ms, conv_fl, conv_by, others = {}, 0.0, 0.0, 0.0, {}
    for nid, typ, nm, t_ms in times:
        if typ != 'Convolution':
            if typ not in ('Const', 'Parameter', 'Reshape'):
                others[typ] = others.get(typ, 0.0) + t_ms
            continue
        node = G.nodes[nid]
        members = [nid] + list(getattr(ex, '_siblings', {}).get(nid, []))
        fl = sum(work[m][0] for m in members)
        # fp16 tensors: the shared input once, every member's weights and output (2 bytes each; conv1's input is the fp32 image)
        xin = int(np.prod(node['input'][0]['dims']))
        by = (4.0 if node['input'][1]['dims'][2] == 7 else 2.0) * xin
        for m in members:
            by += 2.0 * (int(np.prod(G.nodes[m]['input'][1]['dims'])) + int(np.prod(next(iter(G.nodes[m]['output'].values()))['dims'])))
        kh = node['input'][1]['dims'][2]
        fam = '{}x{} {}'.format(kh, kh, node.get('_hip_f16', '?'))
        if nid in getattr(ex, '_pool_conv', {}) and 'MaxPool' not in fam:
            fam = 'MaxPool + ' + fam
        agg = fams.setdefault(fam, {'launches': 0, 'ms': 0.0, 'flops': 0.0, 'bytes': 0.0})
        agg['launches'] += 1; agg['ms'] += t_ms; agg['flops'] += fl; agg['bytes'] += by
        conv_ms += t_ms; conv_fl += fl; conv_by += by
    per_kernel = {}
    for fam, agg in sorted(fams.items(), key=lambda kv: -kv[1]['ms']):
        bound = min(PEAK_MFMA_F16_TFLOPS, agg['flops'] / agg['bytes'] * PEAK_HBM_GBS / 1e3)
        tf = agg['flops'] / (agg['ms'] * 1e-3) / 1e12
        per_kernel[fam] = {'launches': agg['launches'], 'ms': round(agg['ms'], 4), 'TFLOPs': round(tf, 1), 'frac_of_f16_mfma_peak': round(tf / PEAK_MFMA_F16_TFLOPS, 4),
                           'bound_TFLOPs': round(bound, 1), 'frac_of_bound': round(tf / bound, 3), 'GBs_algorithmic': round(agg['bytes'] / (agg['ms'] * 1e-3) / 1e9, 1)}
    n_launch = sum(a_['launches'] for a_ in fams.values())
    tf = conv_fl / (conv_ms * 1e-3) / 1e12
    roof = {'bound': 'mfma', 'achieved': tf, 'peak': PEAK_MFMA_F16_TFLOPS, 'unit': 'TFLOP/s', 'frac': tf / PEAK_MFMA_F16_TFLOPS,
            'kernel': 'all {} Convolution launches of the FP16 pass (v_mfma_f32_32x32x16_f16; algorithmic flops: tile padding not counted)'.format(n_launch),
            'launches_per_pass': n_launch, 'flops_per_launch': conv_fl / n_launch, 'avg_launch_us': conv_ms / n_launch * 1e3,
            'algorithmic_bytes_per_launch': conv_by / n_launch, 'hbm_GBs_algorithmic': round(conv_by / (conv_ms * 1e-3) / 1e9, 1),
            'hbm_frac_algorithmic': round(conv_by / (conv_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 3),
            'per_kernel': per_kernel, 'other_launches_ms': {k: round(v, 4) for k, v in sorted(others.items(), key=lambda kv: -kv[1])},
            'traffic': None, 'traffic_ratio': None, 'traffic_source': None,
            'measured_on': 'one per-layer pass of the FP16 entry (one hipEvent bracket per launch, one stream; a bracket adds ~10 us to launches of 15-300 us)',
            '_conv_ms': conv_ms, '_conv_fl': conv_fl}
    for path in sorted(glob.glob(os.path.join(REPO, 'profiles', '*_fp16_traffic.json')), reverse=True):
        try:
            doc = json.load(open(path))
            k = doc['kernels']['convolution_kernels']
            roof['traffic'] = k['read_bytes_per_launch'] + k['write_bytes_per_launch']
            roof['traffic_ratio'] = round(roof['traffic'] / (conv_by / n_launch), 3)
            roof['traffic_source'] = 'static: {} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/profile_fp16.sh, tree {}; not re-measured in this run)'.format(
                os.path.relpath(path, REPO), doc.get('tree', 'unknown'))
            break
        except Exception:
            continue
    return roof


def fp16_network(blob):
    """GoogLeNet as an FP16 IR read with fp16_as_fp32=False (the FP16 entry of extra_configs), batch 256, and its device-resident feed."""
    import tempfile
    from pyopenvino_amd import IECore, device, synth
    ie = IECore()
    gxml = os.path.join(REPO, 'models', MODEL + '.xml')
    with tempfile.TemporaryDirectory() as tmp:
        xml16, blob16 = synth.fp16_ir(gxml, blob, tmp)
        net = ie.read_network(xml16, weights=blob16, fp16_as_fp32=False)
    assert net.f16_mfma
    net.set_batch(BATCH_PER_GPU)
    ex = ie.load_network(net)
    x = device.DeviceTensor.from_numpy(synth.uniform_pixels(1000, (BATCH_PER_GPU, 3, 224, 224)))
    return net, ex, {net.inputs[0]['name']: x}


def extra_configs(blob=None):
    """BASELINE configs 2 and 5 next to the headline (one synchronous infer() at a time, input resident in HBM, result on the
    host; median of 10-20): mnist batch 64 on the shipped weights, ssd_mobilenet_v1_coco batch 128 (whole IR: prior boxes
    folded, DetectionOutput on the device) on synthetic weights.  Reference numbers for context (CPU, batch 1, hardware
    unstated): integrity_test_expected_result.txt:8 mnist 'special' 8.6 ms, :50 googlenet 0.554 s, :71 ssd 18.26 s per image."""
    from pyopenvino_amd import IECore, device, synth
    out = []
    ie = IECore()
    net = ie.read_network(os.path.join(REPO, 'models', 'mnist.xml'))
    net.set_batch(64)
    ex = ie.load_network(net)
    x = device.DeviceTensor.from_numpy(np.concatenate([synth.uniform_pixels(50 + i, (1, 1, 28, 28)) for i in range(64)], 0))
    rate, ms = median_infer_rate(ex, {net.inputs[0]['name']: x}, 64, 20)
    out.append({'workload': 'models/mnist.xml fp32 batch 64 (shipped weights)', 'images_per_sec': round(rate, 1), 'ms_per_infer': round(ms, 4),
                'reference_cpu_ms_per_image': 8.6})
    del ex, net
    xml = os.path.join(REPO, 'models', 'ssd_mobilenet_v1_coco.xml')
    net = ie.read_network(xml, weights=synth.synth_weights(xml, WEIGHT_SEED))
    net.set_batch(128)
    ex = ie.load_network(net)
    x = device.DeviceTensor.from_numpy(synth.uniform_pixels(9, (128, 3, 300, 300)))
    feed = {net.inputs[0]['name']: x}
    rate, ms = median_infer_rate(ex, feed, 128, 10)
    ex.device_timing, ex.compute_streams = 'all', 1
    ex.infer(feed)
    by_type = {}
    for nid, typ, nm, t_ms in ex.device_times_ms():
        by_type[typ] = by_type.get(typ, 0.0) + t_ms
    ex.device_timing = None
    for host_side in ('Const', 'Parameter', 'Reshape'):        # no launch behind them: their brackets time nothing but themselves
        by_type.pop(host_side, None)
    total = sum(by_type.values())
    top = max(by_type.items(), key=lambda kv: kv[1])
    work = collect_work(net)
    conv_fl = sum(fl for nid, (fl, _) in work.items() if net.G.nodes[nid]['type'] == 'Convolution' and nid not in ex._fused_away)
    out.append({'workload': 'models/ssd_mobilenet_v1_coco.xml fp32 batch 128, whole IR (synthetic weights seed {})'.format(WEIGHT_SEED),
                'images_per_sec': round(rate, 1), 'ms_per_infer': round(ms, 3),
                'dominant_op': top[0], 'dominant_op_fraction_of_device_time': round(top[1] / total, 3),
                'convolution_TFLOPs_algorithmic': round(conv_fl / (by_type.get('Convolution', 1e9) * 1e-3) / 1e12, 1),
                'device_ms_by_op': {k: round(v, 3) for k, v in sorted(by_type.items(), key=lambda kv: -kv[1])[:6]},
                'reference_cpu_ms_per_image': 18260.0})
    del ex, net, x
    if blob is not None:
        # the FP16 IR of the headline model (every constant stored as f16, what Model Optimizer --data_type FP16
        # writes) read with fp16_as_fp32=False: Convolution and MatMul round their operands to fp16 and run on the f16 matrix
        # cores with fp32 accumulation, every tensor stays fp32 in HBM.  A separate entry: never the headline dtype.
        net, ex, feed = fp16_network(blob)
        rate, ms = median_infer_rate(ex, feed, BATCH_PER_GPU, 10)
        fp16_roof = fp16_roofline(ex, net, feed)
        conv_ms, conv_fl = fp16_roof.pop('_conv_ms'), fp16_roof.pop('_conv_fl')
        out.append({'workload': 'models/googlenet-v1.xml as an FP16 IR, batch 256: fp16 operands on the f16 matrix cores (v_mfma_f32_32x32x16_f16), fp32 '
                                'accumulation; the inception modules on fp16 tensors in HBM (channels blocked by eight: module inputs, reduce tensors, '
                                'Concat buffers, and the stem from the output of conv1 on), the classifier fp32; one synchronous infer() at a time',
                    'dtype': 'f16 operands / f32 accumulate', 'images_per_sec': round(rate, 1), 'ms_per_infer': round(ms, 3),
                    'convolution_ms': round(conv_ms, 3), 'convolution_TFLOPs_algorithmic': round(conv_fl / (conv_ms * 1e-3) / 1e12, 1),
                    'roofline': fp16_roof,
                    'note': 'pvhip_conv2d_f16_c8_multi on the inception modules (blocked fp16 in and out: the 1x1 arms as one launch, 3x3 / 5x5, '
                            'MaxPool + pool_proj with the pooling in the operand read; producer waves + LDS-DMA of whole rows, weights from L2: '
                            'within 2x of their matrix-pipe floor; copies, stores and weights fill most of the rest), conv2 on the same kernel, conv1 from row spans of '
                            'the padded image (pvhip_conv2d_f16_stem, blocked fp16 output), MaxPool + LRN and LRN + MaxPool on blocked tensors.  Far '
                            'from the 2.5 PFLOP/s f16 MFMA peak'})
        del ex, net, feed
        # conv1 as Winograd F(3x3,4x4) on the space-to-depth image (opt-in, PVHIP_CONV_STEM_WINO=1): faster than the row-span kernel of the headline, at a
        # lower fraction of the MFMA peak on executed flops and not its bits.  A separate entry, one synchronous infer() at a time, both ways on this box.
        rates = {}
        knob_before = os.environ.get('PVHIP_CONV_STEM_WINO')
        for knob in ('0', '1'):
            os.environ['PVHIP_CONV_STEM_WINO'] = knob
            device.reload_settings()
            net = ie.read_network(os.path.join(REPO, 'models', MODEL + '.xml'), weights=blob)
            net.set_batch(BATCH_PER_GPU)
            ex = ie.load_network(net)
            x = device.DeviceTensor.from_numpy(synth.uniform_pixels(1000, (BATCH_PER_GPU, 3, 224, 224)))
            rates[knob] = median_infer_rate(ex, {net.inputs[0]['name']: x}, BATCH_PER_GPU, 10)
            del ex, net, x
        if knob_before is None:
            os.environ.pop('PVHIP_CONV_STEM_WINO', None)
        else:
            os.environ['PVHIP_CONV_STEM_WINO'] = knob_before
        device.reload_settings()
        out.append({'workload': 'models/googlenet-v1.xml fp32 batch 256 with conv1 as Winograd F(3x3,4x4) on the space-to-depth image (PVHIP_CONV_STEM_WINO=1, opt-in: '
                                'not the headline); one synchronous infer() at a time',
                    'dtype': 'f32', 'images_per_sec': round(rates['1'][0], 1), 'ms_per_infer': round(rates['1'][1], 3),
                    'default_images_per_sec': round(rates['0'][0], 1), 'default_ms_per_infer': round(rates['0'][1], 3),
                    'note': 'pvhip_conv2d_stem_wino_f32: 0.34 of conv1\'s multiplies, ~0.26 of the fp32 MFMA peak on executed flops (the row-span kernel: 0.71-0.735)'})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=BATCH_PER_GPU, help='images per GPU (BASELINE: 256)')
    ap.add_argument('--cpu-images', type=int, default=60, help='images timed on the CPU baseline (0 = skip)')
    ap.add_argument('--no-node-timing', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip extra_configs (mnist batch 64, SSD batch 128)')
    ap.add_argument('--fp16-passes', type=int, default=0, help='profiling hook: run ONLY this many eager passes of the FP16 entry (GoogLeNet as an FP16 IR, batch 256, one stream) and exit')
    ap.add_argument('--min-seconds', type=float, default=1.0, help='repeat the block of --steps steps until this much has been timed')
    ap.add_argument('--streams', type=int, default=0, help='compute streams the scheduler forks branches onto (0 = engine default)')
    ap.add_argument('--requests', type=int, default=int(os.environ.get('PVHIP_BENCH_REQUESTS', '6')), help='(default also from PVHIP_BENCH_REQUESTS: a sweep hook for a driver that cannot pass flags) ''infer requests in flight per GPU (each a whole batch; 1 = synchronous infer(); at most 8).  Six: same box, alternating, '
                    '3 / 4 / 5 / 6 / 7 / 8 requests: 53.3 / 52.3 / 53.4 / 53.8 / 53.5 / 52.8 k images/s -- the persistent kernels of more requests take CUs from each other')
    args = ap.parse_args()

    from pyopenvino_amd import shard
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # not started by a launcher: be one.  N fresh processes of this command, one per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
        # MASTER_* exported as torch.distributed.run would; this process has not touched the GPU and never will.
        sys.exit(shard.launch_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    from pyopenvino_amd import IECore, device, synth
    from pyopenvino_amd.op_plugins import Convolution as conv_plugin
    if args.fp16_passes > 0:
        # scripts/profile_fp16.sh: the FP16 pass under rocprofv3 (kernel trace / PMC passes), eager dispatch on one stream so that a launch's duration is its own
        device.init(0)
        os.environ['PVHIP_AUTO_GRAPH'] = '0'
        net, ex, feed = fp16_network(synth.synth_weights(os.path.join(REPO, 'models', MODEL + '.xml'), WEIGHT_SEED))
        ex.compute_streams = 1
        for _ in range(args.fp16_passes):
            ex.infer(feed)
        print(json.dumps({'fp16_passes': args.fp16_passes, 'batch': BATCH_PER_GPU}))
        return

    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus != world:
        sys.exit('bench.py --gpus {} inside a launcher that exported WORLD_SIZE={}'.format(args.gpus, world))
    group = shard.TcpGroup() if world > 1 else shard.SingleGroup()
    rank = group.rank
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    device.init(local_rank if device.device_count() > 1 else 0)

    xml = os.path.join(REPO, 'models', MODEL + '.xml')
    blob = synth.synth_weights(xml, WEIGHT_SEED)
    ie = IECore()
    net = ie.read_network(xml, weights=blob)
    net.set_batch(args.batch)
    n_req = max(1, args.requests)
    if args.streams > 0:
        os.environ['PVHIP_STREAMS'] = str(args.streams)
    ex = ie.load_network(net, 'GPU', num_requests=n_req)
    n_streams = ex.compute_streams           # per request
    comm = shard.BatchShardComm(group)
    ex.comm = comm
    lo, hi = comm.shard(args.batch * world)      # every rank: the same global batch (its own slice is args.batch images)
    assert hi - lo == args.batch
    # The Result gather goes over RCCL; if the communicator cannot be created on ANY rank (no librccl, no peer access) all
    # ranks agree to gather through the host group instead -- said loudly on stderr and in the JSON line, never silently.
    gather_path, rccl_error = comm.agree_on_gather()
    if gather_path.startswith('host group (RCCL'):
        print('bench.py rank {}: RCCL unavailable, gathering Result tensors through the host group. {}'.format(rank, rccl_error),
              file=sys.stderr, flush=True)
    rccl_ranks = comm.rccl_ranks()      # ncclCommCount: what RCCL itself says the world is (0: no communicator)

    # synthetic input of this rank's shard, resident in HBM before the timed region
    x_host = synth.uniform_pixels(1000 + rank, (args.batch, 3, 224, 224))
    # rows 0-7 of request 0: the eight images the REFERENCE answered at N=1 on these weights (tests/golden/googlenet_rows8.npz,
    # recorded by tests/golden/make_golden.py from /root/reference): the timed path is checked against them below
    golden = np.load(os.path.join(REPO, 'tests', 'golden', 'googlenet_rows8.npz'))
    assert int(golden['weight_seed']) == WEIGHT_SEED
    n_gold = min(len(golden['image_seeds']), args.batch)
    for i in range(n_gold):
        x_host[i] = synth.uniform_pixels(int(golden['image_seeds'][i]), (1, 3, 224, 224))[0]
    x_dev = device.DeviceTensor.from_numpy(x_host)
    x_req = [x_dev] + [device.DeviceTensor.from_numpy(synth.uniform_pixels(1000 + rank + 100 * r, (args.batch, 3, 224, 224)))
                       for r in range(1, n_req)]
    in_name, out_name = net.inputs[0]['name'], net.outputs[0]['name']
    dispatch_s = [0.0]       # host seconds spent dispatching asynchronous passes
    last_result = {}         # request index -> the Result its LAST pass of the timed region returned (gathered over the ranks)

    def pipelined(steps, first=0, on_sample=None):
        """`steps` forward passes with up to n_req whole-batch requests in flight (request i on its own streams and
        its own resident input); every SAMPLE_EVERY-th pass is taken out of the pipeline when on_sample is given."""
        in_flight, out = [], None

        def finish(r):
            last_result[r] = ex.wait(r)[out_name]
            return last_result[r]

        for step in range(first, first + steps):
            if on_sample is not None and step % SAMPLE_EVERY == 0:
                while in_flight:
                    out = finish(in_flight.pop(0))
                out = on_sample()
                continue
            r = step % n_req
            if r in in_flight:
                in_flight.remove(r)
                out = finish(r)
            t_d = time.perf_counter()
            ex.start_async(r, {in_name: x_req[r]})       # (replays the request's recorded pass: one call; PVHIP_AUTO_GRAPH=0: ~100 dispatches)
            dispatch_s[0] += time.perf_counter() - t_d
            in_flight.append(r)
        while in_flight:
            out = finish(in_flight.pop(0))
        return out

    # set-up, like the weight upload: three passes per request bring the device-memory pool to its steady state (a pass
    # allocates its outputs before the previous ones are released), so that no hipMalloc falls into the timed region
    for req in ex.requests:
        for _ in range(3):
            req.infer({in_name: x_req[req.index]})
    informational = rank == 0 and world == 1 and not args.no_node_timing
    per_node = {}
    pcie_ms = single_rate = single_ms = ceiling = graph_ms = sustained = eager_rate = eager_ms = None
    if informational:
        ceiling = copy_ceiling(device)
        # the same step fed from a HOST array (Parameter uploads 154 MB over PCIe from pageable memory, then the forward
        # pass) -- SURVEY 8(d) asks for the end-to-end rate beside the resident one
        ex.device_timing, ex.compute_streams = None, n_streams
        ex.infer({in_name: x_host})
        t1 = time.perf_counter()
        for _ in range(3):
            ex.infer({in_name: x_host})
        pcie_ms = (time.perf_counter() - t1) / 3 * 1e3
        # ONE request at a time, the reference's own metric shape (inference_engine.py:295-321: B / wall time of one
        # exenet.infer()): the inception arms forked onto the engine's default 4 streams, nothing else in flight
        saved = (ex.compute_streams, ex.stream_base)
        ex.compute_streams, ex.stream_base = int(os.environ.get('PVHIP_STREAMS', '4')), 0
        # eager dispatch first (~100 plugin calls per pass), then infer() as it is: after two identical passes with device-resident
        # inputs it records the pass into a hipGraph and replays it with one call
        auto_graph_env = os.environ.get('PVHIP_AUTO_GRAPH')
        os.environ['PVHIP_AUTO_GRAPH'] = '0'
        eager_rate, eager_ms = median_infer_rate(ex, {in_name: x_dev}, args.batch, 11)
        os.environ.pop('PVHIP_AUTO_GRAPH')
        if auto_graph_env is not None:
            os.environ['PVHIP_AUTO_GRAPH'] = auto_graph_env
        single_rate, single_ms = median_infer_rate(ex, {in_name: x_dev}, args.batch, 11, warm=5)
        graph_ms = single_ms if ex.__dict__.get('_graph') is not None else None
        ex.release_graph()
        ex.compute_streams, ex.stream_base = saved
    ex.device_timing_runs = False
    ex.compute_streams = 1
    if informational or (world > 1 and not args.no_node_timing):
        # the per-layer breakdown, one hipEvent bracket per launch, on one stream, untimed.  In a multi-rank run EVERY rank makes these
        # three passes (their Result gathers are collectives); rank 0's times go into the line (`roofline.per_kernel`, `per_op`)
        ex.device_timing = KERNEL_NODES
        spacer = (device.DeviceTensor.empty((256 * 64 * 112 * 112,)), device.DeviceTensor.empty((256 * 64 * 112 * 112,)))
        for _ in range(3):
            # a ~0.3 ms streaming launch in front of the pass: the bracket of the pass's FIRST launch (data/mean) would otherwise begin on
            # an idle GPU and time the host's dispatch latency and the clock ramp with it (it read 0.10-0.25 ms for a 0.05 ms kernel)
            device.select_stream(0)
            device.call('pvhip_relu_f32', ctypes.c_void_p(spacer[0].ptr), ctypes.c_void_p(spacer[1].ptr), spacer[0].size)
            ex.infer({in_name: x_dev})
            for nid, typ, name, ms in ex.device_times_ms():
                per_node.setdefault(nid, [typ, name, 0.0])[2] += ms / 3.0
        ex.device_timing = None
        spacer = None
    ex.compute_streams = n_streams

    # set-up again for request 0, whose network the single-request measurements above used with other stream settings: its
    # recording must exist before the timed region, like the others' (three passes: two eager, the third records)
    for req in (ex.requests if n_req > 1 else []):
        for _ in range(4):
            if req.runner.__dict__.get('_graph') is not None or req.runner.__dict__.get('_auto_graph', {}).get('failed') \
                    or os.environ.get('PVHIP_AUTO_GRAPH', '1') == '0':
                break
            req.infer({in_name: x_req[req.index]})
    replayed_requests = sum(1 for req in ex.requests if req.runner.__dict__.get('_graph') is not None) if n_req > 1 else 0
    out = pipelined(args.warmup) if n_req > 1 else None
    for _ in range(args.warmup if n_req == 1 else 1):
        out = ex.infer({in_name: x_dev})[out_name]
    assert out.shape == (args.batch * world, 1000) and np.isfinite(out).all()

    # A hipEvent bracket costs ~10-15 us of stream time, so inside the timed region only the dominant kernels
    # (the Convolution launches) are bracketed, one bracket per RUN of consecutive Convolution launches and only on every
    # 20th step, which runs alone and on one stream.
    conv_ms, conv_launches, conv_brackets, sampled_steps = 0.0, 0, 0, 0
    host_dispatch = 0.0

    def sampled_step():
        nonlocal host_dispatch, sampled_steps, conv_ms, conv_launches, conv_brackets
        ex.device_timing, ex.device_timing_runs = {'Convolution'}, True
        streams_before, ex.compute_streams = ex.compute_streams, 1     # a kernel's own duration: one stream, nothing else in flight
        res = ex.infer({in_name: x_dev})[out_name]
        ex.compute_streams, ex.device_timing = streams_before, None
        host_dispatch += sum(t[3] for t in ex.last_node_times if t[1] != 'Result')
        sampled_steps += 1
        for nid, typ, name, ms, count in ex.device_times_ms(with_counts=True):
            conv_ms += ms
            conv_launches += count
            conv_brackets += 1
        return res

    def timed_block(instrumented):
        """EXACTLY args.steps steps between barrier + device synchronisation on both sides; max over ranks.  In an instrumented
        block every SAMPLE_EVERY-th step (the first included) is taken out of the request pipeline and runs alone with hipEvent
        brackets around its Convolution launches: that is where roofline.achieved is measured, inside the timed region."""
        nonlocal host_dispatch
        group.barrier()
        device.synchronize()
        t0 = time.perf_counter()
        ev0 = device.Event().record()
        if n_req > 1:
            dispatch_s[0] = 0.0
            pipelined(args.steps, 0, sampled_step if instrumented else None)
            host_dispatch += dispatch_s[0]
        else:
            for step in range(args.steps):
                if instrumented and step % SAMPLE_EVERY == 0:
                    sampled_step()
                else:
                    ex.infer({in_name: x_dev})
                    host_dispatch += sum(t[3] for t in ex.last_node_times if t[1] != 'Result')
        ev1 = device.Event().record()
        device.synchronize()
        group.barrier()
        elapsed = group.allreduce_max(time.perf_counter() - t0)
        return elapsed, ev0.elapsed_ms(ev1)

    # Block 0, 3, 6, ... are instrumented (one step per 20 runs alone, draining the request pipeline: ~2 % of such a block is
    # the measurement itself); the median over all blocks is what is reported, and the instrumented blocks' own median beside it.
    blocks, instrumented_blocks = [], []
    while True:
        inst = (not args.no_node_timing) and len(blocks) % 3 == 0
        blocks.append(timed_block(inst))
        if inst:
            instrumented_blocks.append(blocks[-1][0])
        # every rank takes the same decision: the block times are already the max over ranks
        if sum(b[0] for b in blocks) >= args.min_seconds or len(blocks) >= 200:
            break
    elapsed = statistics.median(b[0] for b in blocks)
    dev_ms = statistics.median(b[1] for b in blocks)
    n_blocks = len(blocks)

    # ---- parity of the path that was just timed (every rank takes part: a gathered Result is a collective) ----
    # The last Result every request returned inside the timed region -- replayed from its own hipGraph with the other requests in
    # flight beside it -- against (a) the reference's recorded N=1 answers for rows 0-7 of request 0 (of EVERY rank: the gathered
    # tensor carries them at rank * batch), 1e-4 in both norms; (b) one eager, synchronous infer() of the same tensor, bit for bit.
    parity = {'checked_requests': 0, 'rows_vs_reference': 0, 'max_norm_error_vs_reference': None, 'worst_element_of_1e-4_allowance': None,
              'replayed_equals_eager_bits': None}
    if n_req == 1:
        last_result[0] = ex.infer({in_name: x_dev})[out_name]
    saved_env = os.environ.get('PVHIP_AUTO_GRAPH')
    os.environ['PVHIP_AUTO_GRAPH'] = '0'
    try:
        bits_ok = True
        for r in sorted(last_result):
            req = ex.requests[r] if n_req > 1 else None
            eager = (req.infer({in_name: x_req[r]}) if req is not None else ex.infer({in_name: x_dev}))[out_name]
            got = np.asarray(last_result[r])
            assert got.shape == (args.batch * world, 1000) and np.isfinite(got).all(), 'request {}: non-finite Result'.format(r)
            if not np.array_equal(got, np.asarray(eager)):
                bits_ok = False
                print('bench.py rank {}: request {}: the replayed Result of the timed region differs from the eager infer() of the same '
                      'tensor in {} elements'.format(rank, r, int((got != np.asarray(eager)).sum())), file=sys.stderr, flush=True)
            parity['checked_requests'] += 1
            if r == 0 and n_gold:
                rows_vs_reference(got, golden['out'], args.batch, world, n_gold, parity)
        parity['replayed_equals_eager_bits'] = bits_ok
    finally:
        os.environ.pop('PVHIP_AUTO_GRAPH')
        if saved_env is not None:
            os.environ['PVHIP_AUTO_GRAPH'] = saved_env
    parity_ok = bool(parity['replayed_equals_eager_bits']) and (not parity['rows_vs_reference'] or (
        parity['max_norm_error_vs_reference'] <= 1e-4 and parity['worst_element_of_1e-4_allowance'] <= 1.0))
    parity['ok'] = parity_ok
    if informational:
        sustained = mfma_ceiling(device)       # (diagnostic library: mapped only now, after the timed region)

    if rank == 0:
        total_images = args.batch * world * args.steps
        result = {
            'metric': 'images/sec googlenet-v1 fp32 @batch256', 'value': total_images / elapsed, 'unit': 'images/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1000.0 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'models/googlenet-v1.xml 1x3x224x224 fp32, batch {} per GPU, synthetic weights seed {}, '
                                   'input resident in HBM, Result copied to host'.format(args.batch, WEIGHT_SEED),
                       'global_batch': args.batch * world,
                       'parallelism': 'batch shard x{} (one process per GPU), all-gather of Result'.format(world),
                       'result_gather': gather_path, 'rccl_ranks': rccl_ranks,
                       # a scaling point only when RCCL itself saw every rank of the job: anything else (one rank, the host-group gather of a rehearsal)
                       # is a functional run, not a measurement of the multi-GPU path
                       'scaling_measured': bool(world > 1 and rccl_ranks == world),
                       'requests_in_flight': n_req, 'compute_streams_per_request': n_streams,
                       'requests_in_flight_note': 'tuned on ONE GPU with an idle host (4 .. 8 swept, round 5: 6 best); on an 8-GPU node eight host processes share '
                                                  'the host -- sweep with --requests N or PVHIP_BENCH_REQUESTS=N (1 .. 8)',
                       'request_dispatch': ('hipGraph replay: each request records its pass once, on its own stream and tensors, and replays it'
                                            if replayed_requests == n_req and n_req > 1 else 'eager (~100 plugin calls per pass)')},
            'timed_blocks': n_blocks, 'timed_region_s': round(sum(b[0] for b in blocks), 3),
            'block_ms_min_median_max': [round(1e3 * min(b[0] for b in blocks), 3), round(1e3 * elapsed, 3), round(1e3 * max(b[0] for b in blocks), 3)],
            'instrumented_blocks': len(instrumented_blocks),
            'ms_per_step_instrumented_blocks': (1e3 * statistics.median(instrumented_blocks) / args.steps) if instrumented_blocks else None,
            'device_ms_per_step': dev_ms / args.steps,
            'host_dispatch_ms_per_step': 1000.0 * host_dispatch / (args.steps * n_blocks),
            'parity_of_timed_path': parity,
        }
        if single_rate is not None:
            result['single_request_images_per_sec'] = single_rate
            result['single_request_ms_per_infer'] = single_ms
            result['single_request_dispatch'] = ('hipGraph replay (infer() records the pass after two identical passes with device-resident inputs)'
                                                 if graph_ms is not None else 'eager')
            result['single_request_eager_images_per_sec'] = eager_rate
            result['single_request_eager_ms_per_infer'] = eager_ms
        if pcie_ms is not None:
            result['host_input_images_per_sec'] = args.batch / (pcie_ms * 1e-3)    # input uploaded from host memory every step
        if ceiling is not None:
            result['hbm_copy_ceiling_GBs'] = ceiling
        roof = None
        if sampled_steps:
            work = collect_work(net)
            G = net.G
            by_type, families, conv_elsewhere = {}, {}, {}
            all_nodes = per_node
            if not all_nodes:   # multi-rank run: no per-layer pass, Convolution work from the graph
                all_nodes = {nid: [G.nodes[nid]['type'], G.nodes[nid]['name'], 0.0]
                             for nid in work if G.nodes[nid]['type'] == 'Convolution' and nid not in ex._fused_away}
            layers = []
            for nid, (typ, name, ms) in sorted(all_nodes.items()):
                fl, by = work.get(nid, (0.0, 0.0))
                family, exec_fl = typ, (fl if typ == 'MatMul' else 0.0)
                if typ == 'Convolution':
                    family, frac = conv_plugin.kernel_kind(G.nodes[nid])
                    exec_fl = fl * frac
                sibs = getattr(ex, '_siblings', {}).get(nid)
                if sibs:                 # convolutions of the same input launched together: all their flops, the input once
                    in_bytes = 4.0 * int(np.prod(G.nodes[nid]['input'][0]['dims']))
                    for sid in sibs:
                        sfl, sby = work.get(sid, (0.0, 0.0))
                        fl, by, exec_fl = fl + sfl, by + sby - in_bytes, exec_fl + sfl
                    name += ' (+{} siblings)'.format(len(sibs))
                    family += ', sibling launch'
                pin = getattr(ex, '_pool_conv', {}).get(nid)
                if pin is not None:      # MaxPool folded into this 1x1 convolution (still one Convolution launch: same flops, same bytes)
                    name = G.nodes[pin[0]]['name'] + ' + ' + name
                    family = 'MaxPool + 1x1 (conv_pool1x1_kernel)'
                pooled = getattr(ex, '_lrn_pool', {}).get(nid)
                if pooled is not None:   # LRN and the MaxPool behind it (or the other order) as one launch: reads the first one's input once, writes the second one's output once
                    typ = family = 'LRN+MaxPool' if typ == 'LRN' else 'MaxPool+LRN'
                    lrn_in = G.nodes[nid]['input'][0]['dims']
                    pool_out = next(iter(G.nodes[pooled]['output'].values()))['dims']
                    by = 4.0 * (int(np.prod(lrn_in)) + int(np.prod(pool_out)))
                    fl = 0.0
                    name += ' + ' + G.nodes[pooled]['name']
                    folded_conv = getattr(ex, '_stem_conv', {}).get(nid)
                    if folded_conv is not None:
                        # ... and the 1x1 convolution behind the LRN in the same launch (pvhip_maxpool_lrn_conv1x1_f32): the launch stays a stream over
                        # the MaxPool's input -- it is listed with the memory-bound launches, against min(MFMA peak, AI x HBM) -- and carries the
                        # convolution's flops, which therefore do NOT count among the Convolution launches of `roofline`
                        typ = family = 'MaxPool+LRN+1x1'
                        cfl, _ = work.get(folded_conv, (0.0, 0.0))
                        cnode = G.nodes[folded_conv]
                        conv_out = next(iter(cnode['output'].values()))['dims']
                        by = 4.0 * (int(np.prod(lrn_in)) + int(np.prod(conv_out)) + int(np.prod(cnode['input'][1]['dims'])))
                        fl = exec_fl = cfl
                        name += ' + ' + cnode['name']
                        conv_elsewhere[cnode['name']] = {'launch': name, 'gflop': round(cfl / 1e9, 3)}
                for table, key in ((by_type, typ), (families, family)):
                    agg = table.setdefault(key, {'ms': 0.0, 'flops': 0.0, 'exec': 0.0, 'bytes': 0.0, 'launches': 0})
                    agg['ms'] += ms
                    agg['flops'] += fl
                    agg['exec'] += exec_fl
                    agg['bytes'] += by
                    agg['launches'] += 1
                row = {'id': nid, 'type': typ, 'kernel': family, 'name': name, 'ms': round(ms, 4), 'gflop': round(fl / 1e9, 3),
                       'gflop_executed': round(exec_fl / 1e9, 3), 'mb': round(by / 1e6, 2)}
                if ms > 0:
                    bound_tf = min(PEAK_MFMA_F32_TFLOPS, (fl / by) * PEAK_HBM_GBS / 1e3) if (fl > 0 and by > 0) else None
                    row.update({'TFLOPs': round(fl / (ms * 1e-3) / 1e12, 2) if fl > 0 else None,
                                'TFLOPs_executed': round(exec_fl / (ms * 1e-3) / 1e12, 2) if fl > 0 else None,
                                'GBs': round(by / (ms * 1e-3) / 1e9, 1),
                                'bound': ('mfma' if bound_tf == PEAK_MFMA_F32_TFLOPS else 'hbm') if fl > 0 else 'hbm',
                                'bound_TFLOPs': round(bound_tf, 1) if bound_tf else None,
                                'frac_of_bound': round((exec_fl / (ms * 1e-3) / 1e12) / bound_tf, 3) if bound_tf else round(by / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 3)})
                layers.append(row)
            conv = by_type.get('Convolution')
            if conv:
                assert conv_launches == conv['launches'] * sampled_steps, (conv_launches, conv['launches'], sampled_steps)
                layer_pass_ms = conv['ms']
                conv['ms'] = conv_ms / sampled_steps             # run brackets inside the timed blocks
                n_launch = conv['launches']
                flops_per_launch = conv['flops'] / n_launch          # algorithmic: 2*N*K*C*kh*kw*oh*ow, averaged
                avg_launch_ms = conv['ms'] / n_launch                # hipEvents on the compute stream, timed steps
                tf = flops_per_launch / (avg_launch_ms * 1e-3) / 1e12
                tf_exec = conv['exec'] / (conv['ms'] * 1e-3) / 1e12
                traffic, traffic_src, traffic_by_kernel = None, None, None
                # (the committed counters are those of the batch-256 workload: no static traffic beside another batch's algorithmic bytes)
                for path in (sorted(glob.glob(os.path.join(REPO, 'profiles', '*_traffic.json')), reverse=True) if args.batch == BATCH_PER_GPU else []):
                    try:
                        doc = json.load(open(path))
                        k = doc['kernels']['convolution_kernels']
                        traffic = k['read_bytes_per_launch'] + k['write_bytes_per_launch']
                        # the same split as scripts/summarize_profile.py::conv_family makes of rocprofv3's kernel names, with the algorithmic
                        # bytes (4 * (input + weights + output), the shared input of a multi-arm launch once) of THIS run's launches beside
                        # the counted ones: traffic_ratio well above 1 = re-reads
                        algo = {}
                        for fam, agg in families.items():
                            if agg['flops'] <= 0 or fam == 'MatMul':
                                continue
                            key = ('conv_pool1x1_kernel' if 'conv_pool1x1' in fam else 'winograd6' if ('F(4x4' in fam or '5x5)' in fam)
                                   else 'conv_wino_kernel' if 'F(2x2,3x3)' in fam else 'conv_stem_kernel' if 'stem' in fam
                                   else 'conv_module_kernel' if 'module launch' in fam else 'conv_pw_kernel' if 'pointwise' in fam
                                   else 'conv_igemm_dma_kernel')
                            a = algo.setdefault(key, [0.0, 0])
                            a[0] += agg['bytes']
                            a[1] += agg['launches']
                        # the six-point layers run on two kernels (conv_wino4s_kernel: shared V; conv_wino4_kernel: two workgroups per CU) and the
                        # per-layer table does not say which: their counted traffic is merged for the ratio
                        fams = dict(doc.get('convolution_kernels_by_family', {}))
                        six = [fams.pop(k) for k in ('conv_wino4s_kernel', 'conv_wino4_kernel') if k in fams]
                        if six:
                            n6 = sum(v.get('launches_per_step') or 0 for v in six) or 1.0
                            fams['winograd6'] = {'launches_per_step': n6,
                                                 'read_bytes_per_launch': sum(v['read_bytes_per_launch'] * (v.get('launches_per_step') or 0) for v in six) / n6,
                                                 'write_bytes_per_launch': sum(v['write_bytes_per_launch'] * (v.get('launches_per_step') or 0) for v in six) / n6}
                        traffic_by_kernel = {}
                        for name, v in fams.items():
                            name = 'conv_wino4s_kernel + conv_wino4_kernel' if name == 'winograd6' else name
                            counted = v['read_bytes_per_launch'] + v['write_bytes_per_launch']
                            a = algo.get('winograd6' if name.startswith('conv_wino4s_kernel +') else name)
                            per_launch = (a[0] / a[1]) if a and a[1] else None
                            traffic_by_kernel[name] = {'launches_per_step': v.get('launches_per_step'), 'bytes_per_launch': counted,
                                                       'read_bytes_per_launch': v['read_bytes_per_launch'], 'write_bytes_per_launch': v['write_bytes_per_launch'],
                                                       'algorithmic_bytes_per_launch': per_launch,
                                                       'launches_per_step_this_run': a[1] if a else None,
                                                       'traffic_ratio': round(counted / per_launch, 3) if per_launch else None}
                        traffic_src = 'static: {} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/profile_bench.sh, tree {}; ' \
                                      'not re-measured in this run)'.format(os.path.relpath(path, REPO), doc.get('tree', 'unknown'))
                        break
                    except Exception:
                        continue
                per_kernel = {}
                for fam, agg in sorted(families.items(), key=lambda kv: -kv[1]['ms']):
                    if agg['flops'] <= 0 or agg['ms'] <= 0:
                        continue
                    bound_tf = min(PEAK_MFMA_F32_TFLOPS, agg['flops'] / agg['bytes'] * PEAK_HBM_GBS / 1e3)
                    per_kernel[fam] = {'launches': agg['launches'], 'ms': round(agg['ms'], 4),
                                       'TFLOPs_algorithmic': round(agg['flops'] / (agg['ms'] * 1e-3) / 1e12, 1),
                                       'TFLOPs_executed': round(agg['exec'] / (agg['ms'] * 1e-3) / 1e12, 1),
                                       'bound_TFLOPs': round(bound_tf, 1),
                                       'frac_executed_of_bound': round(agg['exec'] / (agg['ms'] * 1e-3) / 1e12 / bound_tf, 3)}
                roof = {'bound': 'mfma', 'achieved': tf_exec, 'peak': PEAK_MFMA_F32_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': tf_exec / PEAK_MFMA_F32_TFLOPS,
                        'achieved_algorithmic': tf, 'frac_algorithmic': tf / PEAK_MFMA_F32_TFLOPS,
                        'frac_note': '`achieved` / `frac` count the flops the matrix cores EXECUTE per launch (a roofline fraction: never above 1); '
                                     '`achieved_algorithmic` / `frac_algorithmic` count SURVEY 8(d)\'s algorithmic flops (2*N*K*C*kh*kw*oh*ow), of which the '
                                     'Winograd families execute 16/36 (F(2x2,3x3)), 36/144 (F(4x4,3x3)) or 36/100 (F(2x2,5x5)), padded patches included -- '
                                     'that figure says how fast the layers are done, not how busy the MFMA pipe is, and may exceed 1',
                        'sustained': dict(sustained, frac_of_sustained=round(tf_exec / sustained['TFLOPs'], 3)) if sustained else None,
                        'traffic': traffic, 'traffic_source': traffic_src, 'traffic_by_kernel': traffic_by_kernel or None,
                        'traffic_ratio': round(traffic / (conv['bytes'] / n_launch), 3) if traffic else None,
                        'kernel': 'all Convolution launches of a step: conv_wino4s_kernel / conv_wino4_kernel (F(4x4,3x3), F(2x2,5x5): shared-V form where it pays) + conv_wino_kernel (F(2x2,3x3)) + conv_pw_kernel '
                                  '(1x1; the 1x1 / 3x3_reduce / 5x5_reduce convolutions of an inception module are one launch) + conv_pool1x1_kernel (MaxPool + pool_proj) '
                                  '+ conv_stem_f32_kernel (conv1: row spans of the image, weights resident in registers) + conv_igemm_dma_kernel (whatever else): '
                                  '{} launches per step for {} of the 57 Convolution nodes, bias+ReLU fused{}'.format(
                                      n_launch, 57 - len(conv_elsewhere),
                                      '; ' + ', '.join(sorted(conv_elsewhere)) + ' runs inside the MaxPool + LRN launch in front of it (listed with the memory-bound launches: per_op / per_kernel "MaxPool+LRN+1x1")' if conv_elsewhere else ''),
                        'launches_per_step': n_launch, 'flops_per_launch': flops_per_launch, 'flops_executed_per_launch': conv['exec'] / n_launch,
                        'convolution_nodes_in_other_launches': conv_elsewhere or None,
                        'avg_launch_us': avg_launch_ms * 1e3,
                        'algorithmic_bytes_per_launch': conv['bytes'] / n_launch,
                        'event_sampled_steps': sampled_steps,
                        'measured_on': 'every {}th step of every third timed block (blocks 0, 3, 6, ...; the first step included), run alone on one stream (the other steps keep {} requests in flight: '
                                       'kernels overlap and a launch has no duration of its own)'.format(SAMPLE_EVERY, n_req),
                        'event_brackets_per_step': conv_brackets // sampled_steps,
                        'per_kernel': per_kernel,
                        'per_kernel_measured_on': 'the untimed per-layer pass (one hipEvent bracket per launch, one stream; Convolution total there {:.3f} ms '
                                                  'against {:.3f} ms with run brackets in the timed blocks)'.format(layer_pass_ms, conv['ms'])}
            breakdown = {}
            rocprof_ms = rocprof_per_op() if args.batch == BATCH_PER_GPU else {}          # (the committed trace is the batch-256 workload)
            for typ, agg in sorted(by_type.items(), key=lambda kv: -kv[1]['ms']):
                row = {'launches': agg['launches'], 'ms_per_step': round(agg['ms'], 4)}
                if agg['flops'] > 0:
                    row['TFLOP/s'] = round(agg['flops'] / (agg['ms'] * 1e-3) / 1e12, 2)
                    row['TFLOP/s executed'] = round(agg['exec'] / (agg['ms'] * 1e-3) / 1e12, 2) if agg['exec'] else None
                    row['frac_mfma_peak'] = round(row['TFLOP/s'] / PEAK_MFMA_F32_TFLOPS, 4)
                if agg['ms'] > 0:
                    row['GB/s'] = round(agg['bytes'] / (agg['ms'] * 1e-3) / 1e9, 1)
                    row['frac_hbm_peak'] = round(row['GB/s'] / PEAK_HBM_GBS, 4)
                prof = rocprof_ms.get(typ)
                if prof is not None:
                    # the same launches' durations in the committed rocprofv3 kernel trace (no bracket around them: a bracket adds ~10 us per launch)
                    row['ms_per_step_rocprofv3'] = round(prof, 4)
                    if agg['bytes'] > 0:
                        row['frac_hbm_peak_rocprofv3'] = round(agg['bytes'] / (prof * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
                breakdown[typ] = row
            result['per_op'] = breakdown
            if rocprof_ms:
                result['per_op_rocprofv3_source'] = 'static: profiles/r05_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the one-stream bench command, scripts/profile_bench.sh; average duration x launches per step)'
            print('per-op breakdown (device time per step):', file=sys.stderr)
            for typ, row in breakdown.items():
                print('  {:12s} {}'.format(typ, row), file=sys.stderr)
            out_dir = os.path.join(REPO, 'gpurun_out')
            if os.path.isdir(out_dir):
                with open(os.path.join(out_dir, 'bench_layers.json'), 'w') as f:
                    json.dump({'device': device.device_name(), 'workload': result['config']['workload'],
                               'note': 'per-launch device time from the untimed per-layer pass of bench.py (one hipEvent bracket per launch on one stream; a bracket '
                                       'adds ~10 us); bound = min(157.3 TFLOP/s fp32 MFMA, arithmetic intensity x 8 TB/s); frac_of_bound uses EXECUTED flops',
                               'by_type': breakdown, 'by_kernel_family': roof['per_kernel'] if roof else None, 'layers': layers}, f, indent=1)
        result['roofline'] = roof
        # BASELINE configs 2 and 5 beside the headline, and the CPU restatement of the reference path: rank 0 at N=1 only
        # (the other ranks would wait for it at the exit barrier)
        if world == 1:
            x_req = x_dev = None            # (closures above still name them)
            for req in ex.requests:
                req.runner.release_device_state()
            result['extra_configs'] = extra_configs(blob) if not args.no_extra else None
        result['cpu_baseline'] = cpu_baseline(blob, args.cpu_images) if (args.cpu_images > 0 and world == 1) else None
        print(json.dumps(result), flush=True)

    comm.close()
    if world > 1:
        group.barrier()
        group.close()
    if not parity_ok:
        sys.exit('bench.py rank {}: the timed path disagrees with its references: {}'.format(rank, parity))


if __name__ == '__main__':
    main()
