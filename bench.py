#!/usr/bin/env python3
"""Benchmark of the hot path: images/sec of googlenet-v1 fp32 at batch 256 per GPU (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole per-layer compute() path (Executable_Network.infer) over one batch of
256 synthetic images per GPU, input already resident in HBM, ending with the Result tensor back on the
host (and, for N > 1, an RCCL all-gather of the Result tensors first).  For N > 1 launch one process per
GPU with `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`; torch.distributed
(gloo) is used for the rendezvous / barrier / max-over-ranks only.

Rank 0 prints ONE JSON line (contract in the task statement) carrying
  roofline      for the dominant kernel (conv_igemm_kernel, fp32 MFMA): algorithmic FLOPs of all 57
                Convolution launches of a step / their summed device time measured with hipEvents on the
                compute stream inside the timed steps;
  cpu_baseline  the oracle (CPU restatement of the reference's 'special' path) timed on this host, N=1 per
                image like the reference, on a bounded sample of the same workload.
A per-op-type breakdown (device ms, algorithmic GB/s or TFLOP/s) goes to stderr and, if the directory
exists, to gpurun_out/bench_breakdown.json.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

MODEL = 'googlenet-v1'
BATCH_PER_GPU = 256
WEIGHT_SEED = 1234
PEAK_MFMA_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 MFMA dense peak
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=BATCH_PER_GPU, help='images per GPU (BASELINE: 256)')
    ap.add_argument('--cpu-images', type=int, default=60, help='images timed on the CPU baseline (0 = skip)')
    ap.add_argument('--no-node-timing', action='store_true')
    ap.add_argument('--streams', type=int, default=0, help='compute streams the scheduler forks branches onto (0 = engine default)')
    ap.add_argument('--requests', type=int, default=8, help='infer requests in flight per GPU (each a whole batch; 1 = synchronous infer())')
    args = ap.parse_args()

    from pyopenvino_amd import IECore, device, shard, synth

    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and world != args.gpus:
        sys.exit('bench.py --gpus {} must be launched with torch.distributed.run --nproc-per-node {}'.format(args.gpus, args.gpus))
    group = shard.TorchGroup('gloo') if world > 1 else shard.SingleGroup()
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
    # The Result gather goes over RCCL; if the communicator cannot be created on ANY rank (no librccl, no peer access) all
    # ranks agree to gather through the host group instead -- said loudly on stderr and in the JSON line, never silently.
    gather_path, rccl_error = ('rccl' if comm.use_rccl else ('none (one rank)' if world == 1 else 'host group (PVHIP_NO_RCCL=1)')), ''
    if comm.use_rccl:
        try:
            comm.init_device()
        except Exception as exc:       # noqa: BLE001 -- whatever it is, the other ranks must hear about it
            rccl_error = '{}: {}'.format(type(exc).__name__, exc)
        if group.allreduce_max(1.0 if rccl_error else 0.0) > 0.0:
            comm.use_rccl = False
            gather_path = 'host group (RCCL communicator unavailable{})'.format(': ' + rccl_error if rccl_error else ' on another rank')
            print('bench.py rank {}: RCCL unavailable, gathering Result tensors through the host group. {}'.format(rank, rccl_error),
                  file=sys.stderr, flush=True)

    # synthetic input of this rank's shard, resident in HBM before the timed region
    x_host = synth.uniform_pixels(1000 + rank, (args.batch, 3, 224, 224))
    x_dev = device.DeviceTensor.from_numpy(x_host)
    x_req = [x_dev] + [device.DeviceTensor.from_numpy(synth.uniform_pixels(1000 + rank + 100 * r, (args.batch, 3, 224, 224)))
                       for r in range(1, n_req)]
    in_name, out_name = net.inputs[0]['name'], net.outputs[0]['name']

    SAMPLE_EVERY = 20        # every 20th timed step (the first one included) is taken out of the pipeline and instrumented
    dispatch_s = [0.0]       # host seconds spent dispatching asynchronous passes

    def pipelined(steps, first=0, on_sample=None):
        """`steps` forward passes with up to n_req whole-batch requests in flight (request i on its own streams and
        its own resident input); every SAMPLE_EVERY-th pass is taken out of the pipeline when on_sample is given."""
        in_flight, out = [], None
        for step in range(first, first + steps):
            if on_sample is not None and step % SAMPLE_EVERY == 0:
                while in_flight:
                    out = ex.wait(in_flight.pop(0))[out_name]
                out = on_sample()
                continue
            r = step % n_req
            if r in in_flight:
                in_flight.remove(r)
                out = ex.wait(r)[out_name]
            ex.start_async(r, {in_name: x_req[r]})
            dispatch_s[0] += sum(t[3] for t in ex.requests[r].runner.last_node_times if t[1] != 'Result')
            in_flight.append(r)
        while in_flight:
            out = ex.wait(in_flight.pop(0))[out_name]
        return out

    # set-up, like the weight upload: three passes per request bring the device-memory pool to its steady state (a pass
    # allocates its outputs before the previous ones are released), so that no hipMalloc falls into the timed region
    for req in ex.requests:
        for _ in range(3):
            req.infer({in_name: x_req[req.index]})
    KERNEL_NODES = {'Convolution', 'MatMul', 'MaxPool', 'AvgPool', 'Add', 'Multiply', 'ReLU', 'SoftMax', 'LRN',
                    'Concat', 'Transpose', 'GroupConvolution', 'Clamp', 'Sigmoid'}
    per_node = {}
    # informational, before the warm-up and the timed region (so that those are the last launches of the process,
    # which is what the committed rocprofv3 summary compares with): the same step fed from a HOST array (Parameter
    # uploads 154 MB over PCIe from pageable memory, then the forward pass) -- SURVEY 8(d) asks for the end-to-end
    # rate beside the resident one; then the per-layer breakdown, one bracket per node
    pcie_ms = None
    if rank == 0 and world == 1 and not args.no_node_timing:
        ex.device_timing, ex.compute_streams = None, n_streams
        ex.infer({in_name: x_host})
        t1 = time.perf_counter()
        for _ in range(3):
            ex.infer({in_name: x_host})
        pcie_ms = (time.perf_counter() - t1) / 3 * 1e3
    ex.device_timing_runs = False
    ex.compute_streams = 1
    if not args.no_node_timing and rank == 0 and world == 1:
        ex.device_timing = KERNEL_NODES
        for _ in range(2):
            ex.infer({in_name: x_dev})
            for nid, typ, name, ms in ex.device_times_ms():
                per_node.setdefault(nid, [typ, name, 0.0])[2] += ms / 2.0
        ex.device_timing = None
    ex.compute_streams = n_streams

    out = pipelined(args.warmup) if n_req > 1 else None
    for _ in range(args.warmup if n_req == 1 else 1):
        out = ex.infer({in_name: x_dev})[out_name]
    assert out.shape == (args.batch * world, 1000) and np.isfinite(out).all()

    # A hipEvent bracket costs ~10-15 us of stream time, so inside the timed region only the dominant kernel
    # (the Convolution launches) is bracketed, one bracket per RUN of consecutive Convolution launches (~14 runs
    # of 57 launches per step) and only on every 20th step, which runs alone and on one stream; the per-layer breakdown (one bracket per node, each
    # inflated by its bracket) is taken in an extra, untimed pass afterwards and is informational only.
    conv_ms, conv_launches, conv_brackets = 0.0, 0, 0
    sampled_steps = 0
    host_dispatch = 0.0
    group.barrier()
    device.synchronize()
    t0 = time.perf_counter()
    ev0 = device.Event().record()
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

    if n_req > 1:
        dispatch_s[0] = 0.0
        out = pipelined(args.steps, 0, None if args.no_node_timing else sampled_step)
        host_dispatch += dispatch_s[0]
    else:
        for step in range(args.steps):
            if (not args.no_node_timing) and step % SAMPLE_EVERY == 0:
                out = sampled_step()
            else:
                out = ex.infer({in_name: x_dev})[out_name]
                host_dispatch += sum(t[3] for t in ex.last_node_times if t[1] != 'Result')
    ev1 = device.Event().record()
    device.synchronize()
    group.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = group.allreduce_max(elapsed)
    dev_ms = ev0.elapsed_ms(ev1)
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
                       'result_gather': gather_path,
                       'requests_in_flight': n_req, 'compute_streams_per_request': n_streams},
            'device_ms_per_step': dev_ms / args.steps,
            'host_dispatch_ms_per_step': 1000.0 * host_dispatch / args.steps,
        }
        if pcie_ms is not None:
            result['host_input_images_per_sec'] = args.batch / (pcie_ms * 1e-3)    # input uploaded from host memory every step
        roof = None
        if sampled_steps:
            work = collect_work(net)
            by_type = {}
            all_nodes = per_node
            if not all_nodes:   # multi-rank run: no per-layer pass, Convolution work from the graph
                all_nodes = {nid: [net.G.nodes[nid]['type'], net.G.nodes[nid]['name'], 0.0]
                             for nid in work if net.G.nodes[nid]['type'] == 'Convolution' and nid not in ex._fused_away}
            for nid, (typ, name, ms) in list(all_nodes.items()):
                fl, by = work.get(nid, (0.0, 0.0))
                sibs = getattr(ex, '_siblings', {}).get(nid)
                if sibs:                 # convolutions of the same input launched together: all their flops, the input once
                    in_bytes = 4.0 * int(np.prod(net.G.nodes[nid]['input'][0]['dims']))
                    for sid in sibs:
                        sfl, sby = work.get(sid, (0.0, 0.0))
                        fl, by = fl + sfl, by + sby - in_bytes
                    work[nid] = (fl, by)
                    all_nodes[nid] = [typ, name + ' (+{} siblings)'.format(len(sibs)), ms]
                pin = getattr(ex, '_pool_conv', {}).get(nid)
                if pin is not None:      # MaxPool folded into this 1x1 convolution (still a Convolution launch: same flops, same bytes)
                    all_nodes[nid] = [typ, net.G.nodes[pin[0]]['name'] + ' + ' + name, ms]
                pooled = getattr(ex, '_lrn_pool', {}).get(nid)
                if pooled is not None:   # LRN and the MaxPool behind it as one launch: reads the LRN input once, writes the pooled tensor once
                    typ = 'LRN+MaxPool'
                    lrn_in = net.G.nodes[nid]['input'][0]['dims']
                    pool_out = next(iter(net.G.nodes[pooled]['output'].values()))['dims']
                    by = 4.0 * (int(np.prod(lrn_in)) + int(np.prod(pool_out)))
                    work[nid] = (0.0, by)
                    all_nodes[nid] = [typ, name + ' + ' + net.G.nodes[pooled]['name'], ms]
                agg = by_type.setdefault(typ, {'ms': 0.0, 'flops': 0.0, 'bytes': 0.0, 'launches': 0})
                agg['ms'] += ms
                agg['flops'] += fl
                agg['bytes'] += by
                agg['launches'] += 1
            conv = by_type.get('Convolution')
            if conv:
                assert conv_launches == conv['launches'] * sampled_steps, (conv_launches, conv['launches'], sampled_steps)
                conv['ms'] = conv_ms / sampled_steps             # run brackets inside the timed region
                n_launch = conv['launches']
                flops_per_launch = conv['flops'] / n_launch          # algorithmic: 2*N*K*C*kh*kw*oh*ow, averaged
                avg_launch_ms = conv['ms'] / n_launch                # hipEvents on the compute stream, timed steps
                tf = flops_per_launch / (avg_launch_ms * 1e-3) / 1e12
                traffic, traffic_src = None, None
                for path in sorted(glob.glob(os.path.join(REPO, 'profiles', '*_traffic.json')), reverse=True):
                    try:
                        k = json.load(open(path))['kernels']['conv_igemm_kernel']
                        traffic = k['read_bytes_per_launch'] + k['write_bytes_per_launch']
                        traffic_src = os.path.relpath(path, REPO)
                        break
                    except Exception:
                        continue
                roof = {'bound': 'mfma', 'achieved': tf, 'peak': PEAK_MFMA_F32_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': tf / PEAK_MFMA_F32_TFLOPS, 'traffic': traffic,
                        'kernel': 'conv_wino4_kernel (conv2/3x3, inception 3a / 3b 3x3: Winograd F(4x4,3x3), 36/144 of the algorithmic multiplies) + conv_wino_kernel '
                                  '(the other seven 3x3 layers: F(2x2,3x3), 16/36; conv_wino4_kernel<2>: seven 5x5 layers, F(2x2,5x5), 36/100) + conv_igemm_dma_kernel '
                                  '(conv1, 1x1, the 7x7-sized 5x5; the 1x1 / 3x3_reduce / 5x5_reduce convolutions of an inception module are one launch): '
                                  '{} launches per step for the 57 '
                                  'Convolution nodes, bias+ReLU fused'.format(n_launch),
                        'launches_per_step': n_launch, 'flops_per_launch': flops_per_launch, 'avg_launch_us': avg_launch_ms * 1e3,
                        'algorithmic_bytes_per_launch': conv['bytes'] / n_launch,
                        'traffic_source': traffic_src, 'event_sampled_steps': sampled_steps,
                        'measured_on': 'every 20th timed step (the first included), run alone on one stream (the other steps keep {} requests in flight with the inception '
                                       'arms on {} streams each: kernels overlap and a launch has no duration of its own)'.format(n_req, n_streams),
                        'event_brackets_per_step': conv_brackets // sampled_steps}
            breakdown = {}
            for typ, agg in sorted(by_type.items(), key=lambda kv: -kv[1]['ms']):
                row = {'launches': agg['launches'], 'ms_per_step': round(agg['ms'], 4)}
                if agg['flops'] > 0:
                    row['TFLOP/s'] = round(agg['flops'] / (agg['ms'] * 1e-3) / 1e12, 2)
                    row['frac_mfma_peak'] = round(row['TFLOP/s'] / PEAK_MFMA_F32_TFLOPS, 4)
                if agg['ms'] > 0:
                    row['GB/s'] = round(agg['bytes'] / (agg['ms'] * 1e-3) / 1e9, 1)
                    row['frac_hbm_peak'] = round(row['GB/s'] / PEAK_HBM_GBS, 4)
                breakdown[typ] = row
            layers = [{'id': nid, 'type': typ, 'name': name, 'ms': round(ms, 4),
                       'gflop': round(work.get(nid, (0, 0))[0] / 1e9, 3), 'mb': round(work.get(nid, (0, 0))[1] / 1e6, 2)}
                      for nid, (typ, name, ms) in sorted(all_nodes.items())]
            print('per-op breakdown (device time per step):', file=sys.stderr)
            for typ, row in breakdown.items():
                print('  {:12s} {}'.format(typ, row), file=sys.stderr)
            out_dir = os.path.join(REPO, 'gpurun_out')
            if os.path.isdir(out_dir):
                with open(os.path.join(out_dir, 'bench_breakdown.json'), 'w') as f:
                    json.dump({'by_type': breakdown, 'layers': layers, 'device': device.device_name()}, f, indent=1)
        result['roofline'] = roof
        # the CPU restatement of the reference path, on rank 0 at N=1 only (the other ranks would wait for it at the exit barrier)
        result['cpu_baseline'] = cpu_baseline(blob, args.cpu_images) if (args.cpu_images > 0 and world == 1) else None
        print(json.dumps(result), flush=True)

    comm.close()
    if world > 1:
        group.barrier()
        group.close()


if __name__ == '__main__':
    main()
