"""Where does a forked (multi-stream) hipGraph recording of a pass crash?  (LESSONS.md lesson 30.)

    python scripts/capture_probe.py                 # the driver: every case in a process of its own, a bounded bisect per crashing plan
    python scripts/capture_probe.py <kind> <streams> <K>   # one case: record the first K dispatched tasks of the plan (0 = all)

A case prints ONE line `probe <kind> <streams> <K>: ok nodes=<n>` or dies; the driver preloads scripts/diag/segv_trace.c so that a
crash inside the HIP runtime leaves its native backtrace on stderr.  Nothing here loops on a failing GPU kernel: the crash is a host
segfault inside hipStreamEndCapture and every case is a different recording."""
import os, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def build(kind, streams):
    from pyopenvino_amd import IECore, synth
    ie = IECore()
    if kind.startswith('ssd'):
        xml = os.path.join(REPO, 'models', 'ssd_mobilenet_v1_coco.xml')
        net = ie.read_network(xml, weights=synth.synth_weights(xml, 1234)); shape = (8, 3, 300, 300)
    else:
        xml = os.path.join(REPO, 'models', 'googlenet-v1.xml'); blob = synth.synth_weights(xml, 1234); shape = (8, 3, 224, 224)
        if kind == 'fp16':
            tmp = tempfile.mkdtemp(); xml16, blob16 = synth.fp16_ir(xml, blob, tmp)
            net = ie.read_network(xml16, weights=blob16, fp16_as_fp32=False)
        else:
            net = ie.read_network(xml, weights=blob)
    net.set_batch(8)
    ex = ie.load_network(net)
    if kind == 'fp32-unfused':
        ex.fuse_epilogues = False; ex.plan_fusion()
    ex.compute_streams = streams
    return net, ex, shape


def one_case(kind, streams, k):
    import ctypes
    from pyopenvino_amd import device, synth
    device.init(0)
    net, ex, shape = build(kind, streams)
    x = device.DeviceTensor.from_numpy(synth.uniform_pixels(5, shape))
    name = net.inputs[0]['name']
    os.environ['PVHIP_AUTO_GRAPH'] = '0'
    for _ in range(2):
        ex.infer({name: x})
    G = ex.ienet.G
    full = list(ex.task_list)
    if k > 0:
        dispatched = [t for t in full if t not in ex._fused_away and G.nodes[t]['type'] not in ('Const', 'Parameter')]
        last = dispatched[min(k, len(dispatched)) - 1]
        ex.task_list = full[:full.index(last) + 1]
        ex.task_list = [t for t in ex.task_list if G.nodes[t]['type'] != 'Result']
    plan = ex.plan_streams()
    streams_used = sorted(set(plan[0].values())) if plan else [0]
    for nid, _ in ex.ienet.find_node_by_type('Result'):
        G.nodes[nid]['comm'] = None
        G.nodes[nid]['_async'] = True
    device.select_stream(ex.stream_base)
    device.call('pvhip_graph_begin_capture')
    ex.defer_sync = True
    try:
        ex.run_tasks(False)
    finally:
        ex.defer_sync = False
    device.select_stream(ex.stream_base)
    handle = ctypes.c_void_p()
    print('probe {} {} {}: ending capture, streams used {}'.format(kind, streams, k, streams_used), flush=True)
    device.call('pvhip_graph_end_capture', ctypes.byref(handle))
    print('probe {} {} {}: ok'.format(kind, streams, k), flush=True)
    # leave without replaying: the question is whether the recording can be made
    os._exit(0)


def run(kind, streams, k, preload):
    env = dict(os.environ, PVHIP_GRAPH_VERBOSE='1')
    if preload:
        env['LD_PRELOAD'] = preload
    r = subprocess.run([sys.executable, os.path.abspath(__file__), kind, str(streams), str(k)], capture_output=True, text=True, timeout=240, env=env)
    ok = r.returncode == 0
    tail = [l for l in (r.stdout + r.stderr).splitlines() if 'probe' in l or 'pvhip_graph' in l or 'segv_trace' in l or '.so' in l or 'Error' in l]
    print('--- {} streams={} K={}: rc={} {}'.format(kind, streams, k, r.returncode, 'OK' if ok else 'CRASH/FAIL'), flush=True)
    for l in tail[-40:]:
        print('    ' + l, flush=True)
    return ok


def main():
    so = os.path.join(tempfile.gettempdir(), 'segv_trace.so')
    rc = subprocess.run(['gcc', '-shared', '-fPIC', '-O1', '-o', so, os.path.join(REPO, 'scripts', 'diag', 'segv_trace.c')]).returncode
    preload = so if rc == 0 else None
    budget = 22                                  # cases in all
    for kind, streams, n_tasks in (('fp32-unfused', 3, 323), ('fp16', 4, 173), ('ssd', 4, 284)):
        if budget <= 0:
            break
        budget -= 1
        if run(kind, streams, 0, preload):
            continue
        lo, hi = 1, n_tasks                      # bisect: the shortest prefix of the plan whose recording crashes
        while hi - lo > 1 and budget > 0:
            mid = (lo + hi) // 2
            budget -= 1
            if run(kind, streams, mid, None):
                lo = mid
            else:
                hi = mid
        print('{} on {} streams: the first {} dispatched tasks record, the first {} crash'.format(kind, streams, lo, hi), flush=True)


if __name__ == '__main__':
    if len(sys.argv) >= 4:
        one_case(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
    else:
        main()
