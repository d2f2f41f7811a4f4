"""Which passes survive hipGraph capture?  Each case in a process of its own (a crash inside the HIP runtime cannot be caught)."""
import faulthandler, os, subprocess, sys, tempfile
faulthandler.enable()
sys.path.insert(0, os.getcwd())
CASES = ['fp32-unfused 3', 'fp16 3', 'ssd 3', 'ssd 2']
if len(sys.argv) < 2:
    for c in CASES:
        r = subprocess.run([sys.executable, __file__] + c.split(), capture_output=True, text=True, timeout=300)
        print('{:18s} rc={} {} | {}'.format(c, r.returncode, (r.stdout.strip().splitlines() or [''])[-1], ' / '.join(l for l in r.stderr.splitlines() if 'pvhip_graph' in l or 'Error' in l)[:400]), flush=True)
    sys.exit(0)
import numpy as np
from pyopenvino_amd import IECore, device, synth
REPO = os.getcwd()
kind, streams = sys.argv[1], int(sys.argv[2])
device.init(0)
ie = IECore()
B = 8
if kind.startswith('ssd'):
    xml = os.path.join(REPO, 'models', 'ssd_mobilenet_v1_coco.xml')
    net = ie.read_network(xml, weights=synth.synth_weights(xml, 1234)); shape = (B, 3, 300, 300)
else:
    xml = os.path.join(REPO, 'models', 'googlenet-v1.xml'); blob = synth.synth_weights(xml, 1234); shape = (B, 3, 224, 224)
    if kind == 'fp16':
        tmp = tempfile.mkdtemp(); xml16, blob16 = synth.fp16_ir(xml, blob, tmp)
        net = ie.read_network(xml16, weights=blob16, fp16_as_fp32=False)
    else:
        net = ie.read_network(xml, weights=blob)
net.set_batch(B)
ex = ie.load_network(net)
if kind == 'fp32-unfused':
    ex.fuse_epilogues = False; ex.plan_fusion()
ex.compute_streams = streams
x = device.DeviceTensor.from_numpy(synth.uniform_pixels(5, shape))
name = net.inputs[0]['name']
os.environ['PVHIP_AUTO_GRAPH'] = '0'
want = {k: np.asarray(v) for k, v in ex.infer({name: x}).items()}
if kind == 'ssd-backbone':
    print('skipped'); sys.exit(0)
n_launch = sum(1 for t in ex.last_node_times)
ex.capture_graph({name: x}, streams='plan')
got = ex.infer_graph()
same = all(np.array_equal(np.asarray(got[k]), want[k]) for k in want)
print('captured: {} dispatched nodes, replay bits {}'.format(n_launch, 'same' if same else 'DIFFER'))
