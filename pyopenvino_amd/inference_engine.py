"""OpenVINO-IR loader and list scheduler around the op-plugin boundary.

Same public surface as the reference engine (``pyopenvino/inference_engine.py``): ``IECore``
(``read_network`` ``:74-83``, ``load_network`` ``:86-90``), ``IENetwork`` and ``Executable_Network``
(``schedule_tasks`` ``:218-242``, ``run_tasks`` ``:259-292``, ``infer`` ``:295-321``).  The model is a
``networkx.DiGraph`` whose node dicts carry the IR attributes; a static task list is executed by
calling ``plugins[type].compute(node, inputs, kernel_type=..., debug=False)`` per node -- that call is
the drop-in boundary, and what flows through ``inputs`` here are device-resident tensors.

Differences from the reference, all outside the numeric path:
  * constants are decoded with ``np.frombuffer`` (zero copy) instead of ``struct.unpack`` into tuples;
  * ``IENetwork.set_batch`` rewrites the batch dimension of every activation port so that a batch of
    independent images runs through the same graph (the reference IRs are baked at N=1);
  * plugins are discovered as modules of a package (default ``pyopenvino_amd.op_plugins``) instead of a
    CWD-relative glob;
  * ``infer`` can shard the batch across ranks (one process per GPU): every rank runs the unchanged
    scheduler on its slice and the Result plugin all-gathers the Result tensors.
"""
import importlib
import os
from ctypes import byref, c_void_p as ctypes_void_p
import pkgutil
import sys
import time
import xml.etree.ElementTree as et

import networkx as nx
import numpy as np

from . import common_def

DEFAULT_PLUGIN_PACKAGE = 'pyopenvino_amd.op_plugins'


class Plugins:
    """Registry ``{IR layer type: module}``; a module's file name is the layer type it implements
    (reference inference_engine.py:28-43)."""

    def __init__(self):
        self.plugins = {}

    def import_plugin(self, package: str, module_name: str, plugin_name: str = None):
        module = importlib.import_module(package + '.' + module_name)
        if not hasattr(module, 'compute'):
            return None
        key = plugin_name or module_name
        setattr(self, key, module)
        self.plugins[key] = module
        return module

    def load_plugins(self, plugin_path: str):
        """``plugin_path`` is a dotted package name or a '/'-separated path to one."""
        package = plugin_path.replace(os.sep, '.').replace('/', '.').strip('.')
        pkg = importlib.import_module(package)
        for info in pkgutil.iter_modules(pkg.__path__):
            if not info.name.startswith('_'):
                self.import_plugin(package, info.name)


class IECore:
    def __init__(self, plugin_package: str = DEFAULT_PLUGIN_PACKAGE):
        self.plugins = Plugins()
        self.plugins.load_plugins(plugin_package)

    def construct_node_info(self, net, node_type: str) -> list:
        return [net.G.nodes[node_id] for node_id, _ in net.find_node_by_type(node_type)]

    def check_nodes(self, G: nx.DiGraph):
        missing = {G.nodes[n]['type'] for n in G.nodes} - set(self.plugins.plugins)
        if missing:
            print('Unsupported nodes : {}'.format(sorted(missing)))
        return missing

    def read_network(self, model: str, weights=None, fp16_as_fp32=None):
        """``model``: path of the IR ``.xml``.  ``weights``: path of the ``.bin`` (default: next to the
        xml, as the reference does) or the blob itself (bytes / uint8 ndarray) when the weights were
        synthesised in memory.  ``fp16_as_fp32``: run an FP16 IR with fp32 tensors AND fp32 arithmetic (constants upcast once
        at load, every FP16 port declared FP32); default: what the plugin package asks for (``COMPUTE_FP32``).  False with a
        package that declares ``F16_MFMA`` (this one): Convolution and MatMul round their operands to fp16 and run on the f16
        matrix cores with fp32 accumulation (``net.f16_mfma``), and the tensors between them are fp16 in HBM wherever the plan
        can keep them so (``plan_fusion`` / ``plan_c8_modules``: channels blocked by eight, ``device.BlockedHalf`` -- GoogLeNet
        from conv1's output to pool5; ``PVHIP_CONV_F16_C8=0``: fp32 NCHW tensors holding fp16 values); the ports stay declared
        FP32, which is what a reader outside those kernels gets.  With any other package the IR is left as it is (FP16 ports,
        float16 constants: the reference's own mode)."""
        net = IENetwork(self)
        net.read_IR_Model(model, weights)
        net.parse_IR_XML()
        net.build_graph()
        net.set_constants_to_graph()
        if fp16_as_fp32 is None:
            fp16_as_fp32 = all(getattr(sys.modules.get(m.__package__), 'COMPUTE_FP32', False) for m in self.plugins.plugins.values())
        packages = [sys.modules.get(m.__package__) for m in self.plugins.plugins.values()]
        f16_mfma = fp16_as_fp32 is False and all(getattr(pkg, 'F16_MFMA', False) for pkg in packages)
        if fp16_as_fp32 or f16_mfma:
            net.promote_fp16()
            net.f16_mfma = f16_mfma and net.ir_precision == 'FP16'      # an FP32 IR keeps its fp32 kernels
        net.inputs = self.construct_node_info(net, 'Parameter')
        net.outputs = self.construct_node_info(net, 'Result')
        return net

    def load_network(self, network, device_name: str = 'GPU', num_requests: int = 1):
        """`num_requests` (accepted and ignored by the reference, inference_engine.py:86) is the number of infer
        requests that may be in flight at once: `exenet.requests[i].start_async(inputs)` / `.wait()`, each request
        with its own graph state and its own compute streams.  `exenet.infer()` stays the synchronous call."""
        exenet = Executable_Network(network)
        self.check_nodes(exenet.ienet.G)
        exenet.schedule_tasks()
        exenet.create_requests(max(1, int(num_requests)))
        return exenet


class IENetwork:
    def __init__(self, iecore: IECore):
        self.ie = iecore
        self.xml = None
        self.bin = None
        self.G = None
        self.layers = None
        self.edges = None
        self.inputs = None
        self.outputs = None
        self.batch_size = 1
        self.f16_mfma = False      # FP16 IR read with fp16_as_fp32=False: Convolution / MatMul on the f16 matrix cores

    # ------------------------------------------------------------------ IR reading
    def read_IR_Model(self, model, weights=None):
        stem, _ = os.path.splitext(model)
        xml_file = stem + '.xml'
        if not os.path.isfile(xml_file):
            raise Exception('model {} is not found'.format(model))
        self.xml = et.parse(xml_file)
        if isinstance(weights, (bytes, bytearray, memoryview)):
            self.bin = bytes(weights) if not isinstance(weights, bytes) else weights
        elif isinstance(weights, np.ndarray):
            self.bin = weights.tobytes()
        else:
            bin_file = weights if (isinstance(weights, str) and os.path.isfile(weights)) else stem + '.bin'
            if not os.path.isfile(bin_file):
                raise Exception('model {} is not found'.format(model))
            with open(bin_file, 'rb') as f:
                self.bin = f.read()

    @staticmethod
    def _ports(section):
        ports = {}
        if section is None:
            return None
        for port in section.findall('port'):
            ports[int(port.attrib['id'])] = {
                'precision': port.attrib['precision'],
                'dims': tuple(int(d.text) for d in port.findall('dim')),
            }
        return ports

    def parse_IR_XML(self):
        root = self.xml.getroot()
        if root.tag != 'net':
            raise Exception('Not an OpenVINO IR file')
        layers = {}
        for layer in root.iterfind('./layers/layer'):
            info = {k: v for k, v in layer.attrib.items() if k != 'id'}
            data = layer.find('data')
            if data is not None:
                attrs = dict(data.attrib)
                for key in ('shape', 'stride'):
                    if key in attrs:
                        attrs[key] = common_def.string_to_tuple(attrs[key]) if attrs[key].strip() else ()
                info['data'] = attrs
            for tag in ('input', 'output'):
                ports = self._ports(layer.find(tag))
                if ports is not None:
                    info[tag] = ports
            layers[int(layer.attrib['id'])] = info
        self.layers = layers
        self.edges = [(int(e.attrib['from-layer']), int(e.attrib['from-port']),
                       int(e.attrib['to-layer']), int(e.attrib['to-port']))
                      for e in root.iterfind('./edges/edge')]

    def build_graph(self):
        G = nx.DiGraph()
        for node_id, info in self.layers.items():
            G.add_node(node_id, **info)
        for edge in self.edges:
            G.add_edge(edge[0], edge[2], connection=edge)
        assert nx.is_directed_acyclic_graph(G)
        self.G = G

    def set_constants_to_graph(self):
        """Attach each Const's slice of the ``.bin`` blob (little-endian raw data) to its node."""
        blob = memoryview(self.bin)
        for node_id, _ in self.find_node_by_type('Const'):
            node = self.G.nodes[node_id]
            attrs = node['data']
            offset, size = int(attrs['offset']), int(attrs['size'])
            precision = attrs['element_type'].upper()
            code, width = common_def.format_config[precision]
            if offset + size > len(blob):
                raise Exception('Const {} lies outside the weight blob'.format(node.get('name')))
            values = np.frombuffer(blob[offset:offset + size], dtype=np.dtype('<' + code), count=size // width)
            node['const'] = {'data': values, 'element_info': precision, 'size': size,
                             'decode_info': common_def.format_config[precision]}

    def find_node_by_type(self, type: str) -> list:
        return [(n, self.G.nodes[n]['name']) for n in self.G.nodes if self.G.nodes[n]['type'] == type]

    def promote_fp16(self):
        """SURVEY 8(f)-4, first step: an FP16 IR (ports FP16, f16 constants; the reference runs it in numpy float16,
        common_def.py:13-17) is computed with fp32 tensors -- every FP16 port is declared FP32, f16 constants are upcast
        once here, the Parameter takes fp32.  Results are at least as close to exact arithmetic as the reference's."""
        promoted = False
        for nid in self.G.nodes:
            node = self.G.nodes[nid]
            for tag in ('input', 'output'):
                for port in node.get(tag, {}).values():
                    if port['precision'] == 'FP16':
                        port['precision'] = 'FP32'
                        promoted = True
            attrs = node.get('data')
            if attrs is not None and str(attrs.get('element_type', '')).lower() == 'f16':
                attrs['element_type'] = 'f32'
                promoted = True
                if node['type'] == 'Const':
                    data = np.asarray(node['const']['data'], dtype=np.float16).astype(np.float32)
                    node['const'] = {'data': data, 'element_info': 'F32', 'size': data.nbytes,
                                     'decode_info': common_def.format_config['F32']}
        self.ir_precision = 'FP16' if promoted else 'FP32'

    # ------------------------------------------------------------------ batch
    def set_batch(self, batch: int):
        """Run ``batch`` independent images through the graph: multiply the leading dimension of every
        port that carries activations (anything downstream of a Parameter) by ``batch / current``.
        Constant ports keep their shape; Reshape targets in the shipped IRs use -1 / 0 for the batch
        axis, so they need no edit.  A ShapeOf output is a shape vector, not an activation: nothing that hangs off
        one (the SSD prior-box subgraph) is scaled.  DetectionOutput's records are (1, 1, N * keep, 7)
        (DetectionOutput.py:229-235): its third axis scales."""
        batch = int(batch)
        if batch < 1:
            raise ValueError('batch must be >= 1')
        if batch == self.batch_size:
            return
        if batch % self.batch_size and self.batch_size != 1:
            raise ValueError('set_batch works from the IR batch (call it once, or with a multiple)')
        old = self.batch_size
        G = self.G
        carries = G.copy()
        carries.remove_edges_from([(u, v) for u, v in G.edges if G.nodes[u]['type'] == 'ShapeOf'])
        act_nodes = set()
        for pid, _ in self.find_node_by_type('Parameter'):
            act_nodes.add(pid)
            act_nodes.update(nx.descendants(carries, pid))
        records = set()       # nodes whose tensors are detection records (1, 1, N * keep, 7): DetectionOutput and on
        for did, _ in self.find_node_by_type('DetectionOutput'):
            records.add(did)
            records.update(nx.descendants(G, did))

        def scaled(dims, axis=0):
            if len(dims) <= axis:
                return dims
            return tuple(dims[:axis]) + (dims[axis] // old * batch,) + tuple(dims[axis + 1:])

        for nid in act_nodes:
            node = G.nodes[nid]
            if node['type'] != 'ShapeOf':
                for port in node.get('output', {}).values():
                    port['dims'] = scaled(port['dims'], 2 if nid in records else 0)
            if node['type'] == 'Parameter':
                node['data']['shape'] = scaled(tuple(node['data']['shape']))
            for pred in G.pred[nid]:
                if pred in act_nodes and G.nodes[pred]['type'] != 'ShapeOf':
                    sink_port = G.edges[(pred, nid)]['connection'][3]
                    node['input'][sink_port]['dims'] = scaled(node['input'][sink_port]['dims'], 2 if pred in records else 0)
        self.batch_size = batch


class InferRequest:
    """One inference that can be in flight next to others (the OpenVINO infer-request idea behind the reference's
    unused `num_requests`): it owns a copy of the graph state (node outputs, cached device constants) and a set of
    compute streams, so that the kernels of several requests interleave on the device -- the HBM-bound layers of one
    run beside the matrix-core-bound layers of another, and each fills the other's tails."""

    def __init__(self, owner, runner, index: int):
        self.owner, self.runner, self.index = owner, runner, index      # owner: the network load_network returned
        self._in_flight, self._replayed = False, None

    def start_async(self, inputs: dict):
        if self._in_flight:
            raise RuntimeError('request {} is still in flight: wait() first'.format(self.index))
        ex = self.runner
        G = ex.ienet.G
        # The same device-resident tensors as the last calls: the pass is replayed from this request's own recording (one call
        # instead of ~100 dispatches; every request records its own pass on its own stream and keeps its own tensors, so the
        # replays of several requests run side by side like their eager passes do).
        self._replayed = ex._graph_for(inputs, gathers_later=True)
        if self._replayed is not None:
            ex.launch_graph(self._replayed)
            self._in_flight = True
            return
        by_name = {G.nodes[n]['name']: n for n in G.nodes}
        for node_name, val in inputs.items():
            if node_name in by_name:
                G.nodes[by_name[node_name]]['param'] = val
        for nid, _ in ex.ienet.find_node_by_type('Result'):
            G.nodes[nid]['comm'] = None          # the shards are gathered in wait(): one collective at a time, on stream 0
            G.nodes[nid]['_async'] = True        # keep the result on the device: wait() reads it back
        ex.defer_sync = True
        try:
            ex.run_tasks(False)
        finally:
            ex.defer_sync = False
            for nid, _ in ex.ienet.find_node_by_type('Result'):
                G.nodes[nid].pop('_async', None)
        self._in_flight = True

    def wait(self) -> dict:
        ex = self.runner
        G = ex.ienet.G
        ex.wait_done()
        self._in_flight = False
        out = {}
        comm = self.owner.comm
        replayed, self._replayed = self.__dict__.get('_replayed'), None
        for nid, name in ex.ienet.find_node_by_type('Result'):
            value = replayed['results'][name] if replayed is not None else G.nodes[nid]['result']
            if hasattr(value, 'numpy') and not isinstance(value, np.ndarray):
                from . import device
                # The copy to the host synchronises the stream it is issued on: use one that has nothing else queued
                # (this request's own have drained; another request's have not).
                # With sharded batches the gather and the copy go to the copy stream (index 8), which no request computes on:
                # every rank waits for its requests in the same order, so the collectives of the one communicator
                # are issued in the same order everywhere and never overlap each other.
                gathers = comm is not None and comm.world > 1
                device.select_stream(device.COPY_STREAM if gathers else ex.stream_base)
                if gathers:
                    value = comm.allgather_rows(value)
                value = value.numpy() if hasattr(value, 'numpy') and not isinstance(value, np.ndarray) else np.asarray(value)
                device.select_stream(0)
            elif comm is not None and comm.world > 1:
                value = comm.allgather_rows(value)
            G.nodes[nid]['result'] = value
            out[name] = value
        return out

    def infer(self, inputs: dict) -> dict:
        self.start_async(inputs)
        return self.wait()


class CaptureStreamModel:
    """What ROCm 7.2's runtime keeps per stream while a multi-stream hipGraph recording is open, restated from the disassembly of
    its hipStreamWaitEvent / Stream::EndCapture (libamdhip64.so.7.2.70200 +0x2f63b9 / +0x2df7f0; profiles/r04_capture.md).

    When a stream W that is not the origin of the capture waits for an event recorded on stream E, and E's current parent is not
    W, the runtime sets parent(W) = E and appends W to E's list of parallel streams (once) -- on EVERY such wait, not only on the
    one that makes W join.  hipStreamEndCapture then walks those lists recursively from the origin and clears them on the way
    back.  The parent test stops a 2-cycle only while E's parent still IS W; after E has waited for a third stream in between,
    W <-> E (or a longer ring) closes, the walk never returns and the process dies of stack overflow inside hipStreamEndCapture
    (what LESSONS.md lesson 30 filed as "crashes inside the runtime").  The origin never registers anywhere, so a dependency that
    would close a ring is RELAYED through it: the origin waits for E's event, records a fresh one, W waits for that."""

    def __init__(self):
        self.parent = {}        # non-origin stream -> stream of the event it last registered under
        self.lists = {}         # stream -> streams in its parallel-capture list

    def _reaches(self, src, dst):
        todo, seen = [src], set()
        while todo:
            cur = todo.pop()
            if cur == dst:
                return True
            if cur not in seen:
                seen.add(cur)
                todo.extend(self.lists.get(cur, ()))
        return False

    def wait(self, waiter: int, event_stream: int) -> str:
        """Stream `waiter` is about to wait for an event recorded on `event_stream` (0 = the origin).  'plain': issue the wait;
        'relay': it would close a ring in the runtime's lists -- go through the origin (the bookkeeping of the relay's own two
        waits is applied here)."""
        if waiter == 0 or waiter == event_stream:
            return 'plain'                              # the origin registers nowhere
        if self.parent.get(event_stream) == waiter:
            return 'plain'                              # the runtime's own test: nothing is registered
        if event_stream != 0 and self._reaches(waiter, event_stream):
            self.parent[waiter] = 0                     # relayed: origin waits (registers nothing), waiter waits for the origin's event
            self.lists.setdefault(0, set()).add(waiter)
            return 'relay'
        self.parent[waiter] = event_stream
        self.lists.setdefault(event_stream, set()).add(waiter)
        return 'plain'

    def has_ring(self) -> bool:
        return any(self._reaches(w, s) for s, ws in self.lists.items() for w in ws)


class Executable_Network:
    def __init__(self, ienetwork: IENetwork):
        self.ienet = ienetwork
        self.kernel_type = 'hip'        # the reference's 'naive' / 'numpy' / 'special' are accepted too
        self.expected_result = None     # {node name: [precision, dims, ndarray]} (the reference's format) or {node name: ndarray}: per-layer compare hook (cf. :284-287)
        self.expected_rtol = 1.0        # the reference compares with np.allclose(rtol=1) (common_def.py:72)
        self.pickle_node_args = []      # node ids whose (node, inputs) run_tasks dumps to node_args_<id>.pickle (cf. :216, :275-278)
        self.pickle_dir = '.'           # where (the reference writes into the working directory)
        self.task_list = []
        self.last_node_times = []       # [(node id, type, name, host seconds)] of the last run_tasks
        self.comm = None                # shard.BatchShardComm when the batch is sharded over ranks
        self.device_timing = None       # None, 'all', or a set of layer types: bracket those nodes with hipEvents
        self.device_timing_runs = False # True: consecutive bracketed nodes share ONE bracket (a bracket costs ~10-15 us
                                        # of stream time, which a per-node bracket would add to every launch)
        self.fuse_epilogues = True      # run Convolution -> Add(per-channel const) -> ReLU chains as one launch
        self._fusion = {}               # conv node id -> {'bias': const id, 'add': id, 'relu': id or None}
        self._fused_away = set()        # node ids whose compute() is folded into their producer
        self._concat_direct = {}        # Concat node id -> total channels, when every input is written in place
        self._lrn_pool = {}             # LRN node id -> id of the MaxPool folded into it, or MaxPool id -> id of the LRN folded into it
        self._siblings = {}             # Convolution node id -> ids of the convolutions of the same input launched with it
        self._pool_conv = {}            # Convolution node id -> (MaxPool node folded into its input tile, id of the MaxPool's data input)
        self._pre_add = {}              # Convolution node id -> (Add node folded into its padding pass, Const id, id of the Add's data input)
        self.fuse_siblings = os.environ.get('PVHIP_FUSE_SIBLINGS', '1') != '0'
        self._infer_serial = 0
        self._timed = []                # [(node id, type, name, start Event, stop Event)] of the last run_tasks
        # Independent branches of the graph (the four arms of an inception module) go to separate compute
        # streams, ordered by untimed events; only for plugin sets whose tensors live on the device.
        self.compute_streams = int(os.environ.get('PVHIP_STREAMS', '4'))
        self.stream_base = 0            # first compute stream of this network (several requests in flight use disjoint sets)
        self.defer_sync = False         # True: run_tasks returns after the device-side join; the caller waits
        self._stream_plans = {}

    def create_requests(self, count: int):
        """Request 0 runs on this network's own graph; the others on copies of it made now, before anything has been
        uploaded (constants are uploaded and weights packed per request: the IRs' weights are tens of MB).  The
        compute streams (`compute_streams`, 4: what the device's hardware queues take without multiplexing -- more
        streams than that serialise behind each other's event waits) are split evenly between the requests."""
        import copy
        if count > 8:
            raise ValueError('at most 8 requests (one compute stream each)')
        # A graph that has already been loaded or inferred holds device tensors (cached constants, packed weights, node
        # outputs): a deep copy would alias their blocks, and both owners would free them.  Strip that state first: every
        # request uploads and packs its own.
        self.release_device_state()
        per = max(1, int(self.compute_streams) // count)
        self.requests = [InferRequest(self, self, 0)]
        for i in range(1, count):
            twin = IENetwork(self.ienet.ie)
            for key, val in self.ienet.__dict__.items():
                if key not in ('ie', 'G'):
                    twin.__dict__[key] = val
            twin.G = copy.deepcopy(self.ienet.G)
            twin.inputs = self.ienet.ie.construct_node_info(twin, 'Parameter')
            twin.outputs = self.ienet.ie.construct_node_info(twin, 'Result')
            runner = Executable_Network(twin)
            runner.fuse_epilogues = self.fuse_epilogues
            runner.schedule_tasks()
            runner.requests = []
            self.requests.append(InferRequest(self, runner, i))
        if count > 1:
            for i, req in enumerate(self.requests):
                req.runner.stream_base = i * per
                req.runner.compute_streams = per

    def release_device_state(self):
        """Drop every device tensor the graph holds (cached constants, packed weights, node outputs, results); the next
        infer uploads and packs again.  A captured hipGraph holds raw addresses of exactly these tensors (packed weights, cached
        constants, Concat buffers): it goes first, or a later infer_graph() would replay kernels over freed or reused pool blocks."""
        self.release_graph()
        G = self.ienet.G
        for nid in G.nodes:
            node = G.nodes[nid]
            for key in [k for k in node if isinstance(k, str) and k.startswith('_hip_')]:
                del node[key]
            for port in node.get('output', {}).values():
                port.pop('data', None)
            for key in ('result', 'param', '_sibling_out', '_fuse_bias', '_out_into', '_out_c8', '_siblings', '_fuse_pool', '_fuse_pool_in', '_fuse_lrn'):
                node.pop(key, None)

    def start_async(self, request_id: int, inputs: dict):
        self.requests[request_id].start_async(inputs)

    def wait(self, request_id: int) -> dict:
        return self.requests[request_id].wait()

    def schedule_tasks(self):
        """Static list schedule: sources (Const, Parameter) first, then repeated sweeps in node order
        appending every node whose predecessors are all scheduled (same policy as :218-242)."""
        G = self.ienet.G
        order, done = [], set()
        pending = []
        for node_id in G.nodes:
            if G.nodes[node_id]['type'] in ('Const', 'Parameter'):
                order.append(node_id)
                done.add(node_id)
            else:
                pending.append(node_id)
        while pending:
            still = []
            for node_id in pending:
                if all(p in done for p in G.pred[node_id]):
                    order.append(node_id)
                    done.add(node_id)
                else:
                    still.append(node_id)
            if len(still) == len(pending):
                raise RuntimeError('graph has nodes that can never become ready')
            pending = still
        self.list_schedule = list(order)         # the reference's order; plan_fusion may move mutually independent arms (order_for_locality)
        self.task_list = order
        self.plan_fusion()

    def plan_fusion(self):
        self._plan_serial = self.__dict__.get('_plan_serial', 0) + 1      # a captured pass is a pass of ONE plan
        """Peephole over the scheduled graph (SURVEY 8(f)-1): a Convolution whose only consumer is an Add of a
        per-output-channel Const (1,K,1,1), optionally followed by a ReLU as the Add's only consumer, is run
        as ONE launch: the Convolution plugin receives the bias tensor / relu flag on its node dict and
        applies them in the kernel epilogue (the same fp32 add and the same select, so the fused result is
        bit-identical to the three launches); the Add and ReLU nodes are not dispatched and their output
        ports alias the fused tensor.  Plugins that do not understand the hints (any foreign Convolution
        plugin) never see them because fusion is only planned for this package's plugin."""
        self._fusion, self._fused_away, self._lrn_pool, self._siblings, self._pool_conv = {}, set(), {}, {}, {}
        self._pre_add, self._c8_out, self._c8_concat, self._c8_entry = {}, set(), set(), set()
        self._stem_conv = {}                 # MaxPool (leading MaxPool + LRN) -> the 1x1 convolution behind the LRN that rides in the same launch
        if 'list_schedule' in self.__dict__:
            self.task_list = list(self.list_schedule)
        if not self.fuse_epilogues:
            return
        from . import device           # settings only (parsed from the environment at import / reload_settings(): the SAME values the plugins read)
        G = self.ienet.G
        # LRN whose only consumer is a MaxPool the fused kernel covers: one launch, the LRN tensor is never written
        lrn_plugin = self.ienet.ie.plugins.plugins.get('LRN')
        if lrn_plugin is not None and getattr(lrn_plugin, 'SUPPORTS_FUSED_POOL', False):
            for lid in G.nodes:
                if G.nodes[lid]['type'] != 'LRN':
                    continue
                succ = list(G.successors(lid))
                if len(succ) != 1 or G.nodes[succ[0]]['type'] != 'MaxPool' or G.edges[(lid, succ[0])]['connection'][3] != 0:
                    continue
                if lrn_plugin.pool_fusable(G.nodes[lid], G.nodes[succ[0]]):
                    self._lrn_pool[lid] = succ[0]
                    self._fused_away.add(succ[0])
        # the other order: a MaxPool whose only consumer is an LRN (GoogLeNet: pool1/3x3_s2 -> pool1/norm1)
        pool_plugin = self.ienet.ie.plugins.plugins.get('MaxPool')
        if pool_plugin is not None and getattr(pool_plugin, 'SUPPORTS_FUSED_LRN', False):
            for pid in G.nodes:
                if G.nodes[pid]['type'] != 'MaxPool' or pid in self._fused_away:
                    continue
                succ = list(G.successors(pid))
                if len(succ) != 1 or G.nodes[succ[0]]['type'] != 'LRN' or G.edges[(pid, succ[0])]['connection'][3] != 0:
                    continue
                if succ[0] in self._lrn_pool or succ[0] in self._fused_away:
                    continue                 # that LRN already leads an LRN -> MaxPool launch
                if pool_plugin.lrn_fusable(G.nodes[pid], G.nodes[succ[0]]):
                    self._lrn_pool[pid] = succ[0]
                    self._fused_away.add(succ[0])
        conv_plugin = self.ienet.ie.plugins.plugins.get('Convolution')
        if conv_plugin is None or not getattr(conv_plugin, 'SUPPORTS_FUSED_EPILOGUE', False):
            return
        fusable = {t for t in ('Convolution', 'GroupConvolution')
                   if getattr(self.ienet.ie.plugins.plugins.get(t), 'SUPPORTS_FUSED_EPILOGUE', False)}
        for cid in G.nodes:
            if G.nodes[cid]['type'] not in fusable:
                continue
            succ = list(G.successors(cid))
            if len(succ) != 1 or G.nodes[succ[0]]['type'] != 'Add':
                continue
            aid = succ[0]
            if G.edges[(cid, aid)]['connection'][3] != 0:
                continue
            others = [p for p in G.pred[aid] if p != cid]
            if len(others) != 1 or G.nodes[others[0]]['type'] != 'Const':
                continue
            bid = others[0]
            k_out = next(iter(G.nodes[cid]['output'].values()))['dims'][1]
            if tuple(G.nodes[bid]['data']['shape']) != (1, k_out, 1, 1) or G.nodes[bid]['data']['element_type'] != 'f32':
                continue
            rid, act = None, None
            asucc = list(G.successors(aid))
            if len(asucc) == 1 and G.nodes[asucc[0]]['type'] == 'ReLU':
                rid, act = asucc[0], ('relu',)
            elif len(asucc) == 1 and G.nodes[asucc[0]]['type'] == 'Clamp':
                rid = asucc[0]
                act = ('clamp', float(G.nodes[rid]['data']['min']), float(G.nodes[rid]['data']['max']))
            self._fusion[cid] = {'bias': bid, 'add': aid, 'relu': rid, 'act': act, 'into': None}
            self._fused_away.add(aid)
            if rid is not None:
                self._fused_away.add(rid)
        # Second peephole: a channel Concat (axis 1, NCHW) all of whose inputs are ends of fused convolution
        # chains with no other consumer is not dispatched either: each producing convolution writes its
        # channels straight into the Concat's output tensor (Concat.py:9-13 becomes a store pattern).
        self._concat_direct = {}
        tail_of = {}
        for cid, f in self._fusion.items():
            tail_of[f['relu'] if f['relu'] is not None else f['add']] = cid
        for nid in G.nodes:
            node = G.nodes[nid]
            if node['type'] != 'Concat' or int(node['data']['axis']) != 1:
                continue
            out_dims = next(iter(node['output'].values()))['dims']
            preds = list(G.pred[nid])
            if len(out_dims) != 4 or len(preds) < 2 or len(preds) != len(node['input']):
                continue
            plan, coff, ok = [], 0, True
            for pred in preds:                       # edge order == np.concatenate order (Concat.py:11-12)
                cid = tail_of.get(pred)
                if cid is None or len(list(G.successors(pred))) != 1:
                    ok = False
                    break
                k = next(iter(G.nodes[cid]['output'].values()))['dims'][1]
                plan.append((cid, coff))
                coff += k
            if not ok or coff != out_dims[1]:
                continue
            for cid, off in plan:
                self._fusion[cid]['into'] = (nid, off)
            self._concat_direct[nid] = coff
            self._fused_away.add(nid)

        f16 = bool(getattr(self.ienet, 'f16_mfma', False))     # the f16-MFMA kernel fuses the epilogue and the Concat store only
        # A 3x3 / stride 1 / pad 1 MaxPool whose only consumer is a fused 1x1 convolution (pool -> pool_proj): the MaxPool is not
        # dispatched, the convolution reads the MaxPool's input and pools while it builds its input tile.
        if (not f16 or device.conv_f16_dma) and getattr(conv_plugin, 'SUPPORTS_POOLED_INPUT', False) and device.fuse_poolconv != 0:
            for cid in list(self._fusion):
                if G.nodes[cid]['type'] != 'Convolution':
                    continue
                src = next((p_ for p_ in G.pred[cid] if G.edges[(p_, cid)]['connection'][3] == 0), None)
                if src is None or G.nodes[src]['type'] != 'MaxPool' or src in self._fused_away or len(list(G.successors(src))) != 1:
                    continue
                psrc = next((p_ for p_ in G.pred[src] if G.edges[(p_, src)]['connection'][3] == 0), None)
                if psrc is not None and conv_plugin.pooled_fusable(G.nodes[cid], G.nodes[src]):
                    self._pool_conv[cid] = (src, psrc)
                    self._fused_away.add(src)
        # An Add of a per-INPUT-channel Const whose only consumer is a convolution that pads its input in a pass of its own (GoogLeNet:
        # data/mean -> conv1): the padding pass adds the constant on the way and the Add is not dispatched (same fp32 add: same bits).
        if getattr(conv_plugin, 'SUPPORTS_PRE_ADD', False):
            for cid in G.nodes:
                if G.nodes[cid]['type'] != 'Convolution' or cid in self._pool_conv:
                    continue
                src = next((p_ for p_ in G.pred[cid] if G.edges[(p_, cid)]['connection'][3] == 0), None)
                if src is None or G.nodes[src]['type'] != 'Add' or src in self._fused_away or len(list(G.successors(src))) != 1:
                    continue
                preds = list(G.pred[src])
                consts = [p_ for p_ in preds if G.nodes[p_]['type'] == 'Const']
                others = [p_ for p_ in preds if G.nodes[p_]['type'] != 'Const']
                if len(preds) != 2 or len(consts) != 1 or len(others) != 1:
                    continue
                if (conv_plugin.pre_add_fusable(G.nodes[cid], G.nodes[src], G.nodes[consts[0]], True) if f16
                        else conv_plugin.pre_add_fusable(G.nodes[cid], G.nodes[src], G.nodes[consts[0]])):
                    self._pre_add[cid] = (src, consts[0], others[0])
                    self._fused_away.add(src)
        # Third peephole: fused convolution chains that read the SAME tensor with the same geometry and activation (the
        # 1x1, 3x3_reduce and 5x5_reduce arms of an inception module) are one launch of the first of them in schedule
        # order: the input is read once and every output-channel tile stores into the tensor of its own convolution
        # (Convolution.launch_siblings; each output has the bits of its own launch).
        if (not f16 or device.conv_f16_dma) and self.fuse_siblings and getattr(conv_plugin, 'SUPPORTS_SIBLINGS', False):
            position = {t: i for i, t in enumerate(self.task_list)}
            groups = {}
            for cid, f in self._fusion.items():
                if G.nodes[cid]['type'] != 'Convolution':
                    continue
                src = next((G.edges[(p_, cid)]['connection'][:2] for p_ in G.pred[cid] if G.edges[(p_, cid)]['connection'][3] == 0), None)
                if src is None or G.nodes[src[0]]['type'] == 'Const':
                    continue
                geometry = tuple(G.nodes[cid]['input'][1]['dims'][2:]) + tuple(G.nodes[cid]['data'].get(k_) for k_ in ('strides', 'pads_begin', 'pads_end', 'auto_pad'))
                groups.setdefault((tuple(src), f['act'], geometry), []).append(cid)
            for members in groups.values():
                members.sort(key=position.get)
                members = members[:6]
                if len(members) >= 2 and conv_plugin.siblings_fusable([G.nodes[m] for m in members]):
                    self._siblings[members[0]] = members[1:]
                    self._fused_away.update(members[1:])
        # MaxPool -> LRN (one launch already) whose only reader is a fused 1x1 convolution chain that stands alone (no siblings, no Concat slot):
        # the convolution rides in that launch too (GoogLeNet: pool1/3x3_s2 -> pool1/norm1 -> conv2/3x3_reduce) -- the normalised tensor never
        # exists; the MaxPool task returns the convolution's output and the chain's ports alias it.  fp32 IRs only.
        if not f16 and device.fuse_stem_conv != 0 and pool_plugin is not None and getattr(pool_plugin, 'SUPPORTS_FUSED_LRN_CONV', False):
            for pid, lid in self._lrn_pool.items():
                if G.nodes[pid]['type'] != 'MaxPool':
                    continue
                readers = list(G.successors(lid))
                if len(readers) != 1 or G.edges[(lid, readers[0])]['connection'][3] != 0:
                    continue
                cid = readers[0]
                f = self._fusion.get(cid)
                if f is None or G.nodes[cid]['type'] != 'Convolution' or f['into'] is not None or cid in self._siblings or cid in self._fused_away \
                        or cid in self._pool_conv or cid in self._pre_add:
                    continue
                if pool_plugin.lrn_conv_fusable(G.nodes[pid], G.nodes[lid], G.nodes[cid]):
                    self._stem_conv[pid] = cid
                    self._fused_away.update(n_ for n_ in (cid, f['add'], f['relu']) if n_ is not None)
        # FP16 IRs on the f16 matrix cores: a fused 1x1 convolution chain whose ONLY reader is a 3x3 / 5x5 convolution that
        # pvhip_conv2d_f16_c8 covers (3x3_reduce -> 3x3, 5x5_reduce -> 5x5) hands its output over as fp16 with the channels blocked by
        # eight (device.BlockedHalf): what the reference holds there is a float16 tensor too (common_def.py:13-17), and the blocked
        # form is the reader's MFMA operand as it stands.  PVHIP_CONV_F16_C8=0: fp32 NCHW everywhere, as before.
        self._c8_out = set()
        if f16 and device.conv_f16_c8 != 0 and device.conv_f16_dma and getattr(conv_plugin, 'SUPPORTS_C8', False):
            for cid, f in self._fusion.items():
                if G.nodes[cid]['type'] != 'Convolution' or f['into'] is not None or cid in self._pool_conv or cid in self._pre_add:
                    continue
                if f['act'] is not None and f['act'][0] != 'relu':
                    continue
                tail = f['relu'] if f['relu'] is not None else f['add']
                readers = list(G.successors(tail))
                if len(readers) != 1 or G.nodes[readers[0]]['type'] != 'Convolution' or G.edges[(tail, readers[0])]['connection'][3] != 0:
                    continue
                rid = readers[0]
                if rid in self._pool_conv or rid in self._pre_add or rid in self._siblings or rid in self._fused_away:
                    continue
                if conv_plugin.c8_writer_ok(G.nodes[cid]) and conv_plugin.c8_reader_ok(G.nodes[rid]):
                    self._c8_out.add(cid)
            # ... and the stem: a convolution whose only reader is a 3x3 MaxPool (with its LRN folded in: GoogLeNet's conv1 -> pool1 -> norm1)
            # whose readers are convolutions that take a blocked input: the MaxPool plugin pools a blocked tensor as it is
            if device.conv_f16_c8 == 2 and getattr(conv_plugin, 'SUPPORTS_C8_MODULES', False):
                for cid, f in self._fusion.items():
                    if G.nodes[cid]['type'] != 'Convolution' or f['into'] is not None or cid in self._pool_conv or cid in self._siblings or cid in self._fused_away:
                        continue
                    if f['act'] is not None and f['act'][0] != 'relu':
                        continue
                    tail = f['relu'] if f['relu'] is not None else f['add']
                    readers = list(G.successors(tail))
                    if len(readers) != 1 or G.nodes[readers[0]]['type'] not in ('MaxPool', 'LRN') or readers[0] in self._fused_away:
                        continue
                    pid = readers[0]                                 # a 3x3 MaxPool (alone, or leading MaxPool + LRN), or an LRN leading LRN + MaxPool
                    folded = self._lrn_pool.get(pid)
                    if G.nodes[pid]['type'] == 'LRN' and folded is None:
                        continue
                    # the SAME predicates the MaxPool / LRN plugins decide with at run time (blocked_ok): what is planned blocked is blocked
                    if G.nodes[pid]['type'] == 'MaxPool':
                        if not pool_plugin.blocked_ok(G.nodes[pid], G.nodes[folded] if folded is not None else None):
                            continue
                    elif not lrn_plugin.blocked_ok(G.nodes[pid], G.nodes[folded]):
                        continue
                    out_node = folded if folded is not None else pid   # the node folded into the leading one carries the tensor
                    after = list(G.successors(out_node))
                    folded_pools = {p[0] for p in self._pool_conv.values()}

                    def takes_blocked(r):
                        if r in folded_pools:                        # a MaxPool folded into its pool_proj convolution
                            pc = next(c_ for c_, p_ in self._pool_conv.items() if p_[0] == r)
                            return conv_plugin.c8_module_member_ok(G.nodes[pc], G.nodes[r])
                        return r in self._fusion and G.nodes[r]['type'] == 'Convolution' and conv_plugin.c8_module_member_ok(G.nodes[r]) and \
                            (r in self._c8_out or r in self._siblings)
                    if not after or not all(takes_blocked(r) for r in after):
                        continue
                        # the writer: the f16 1x1 launch, the f16 form of the LDS-DMA kernel (conv1), or -- its own input being blocked -- the module form
                    own_src = next((p_ for p_ in G.pred[cid] if G.edges[(p_, cid)]['connection'][3] == 0), None)
                    own_blocked = any(own_src == (self._fusion[c_]['relu'] if self._fusion[c_]['relu'] is not None else self._fusion[c_]['add'])
                                      for c_ in self._c8_out)
                    if conv_plugin.c8_writer_ok(G.nodes[cid]) or conv_plugin.c8_dma_writer_ok(G.nodes[cid]) or \
                            (own_blocked and conv_plugin.c8_module_member_ok(G.nodes[cid])):
                        self._c8_out.add(cid)
            if device.conv_f16_c8 == 2 and device.fuse_stem_conv != 0 and pool_plugin is not None and getattr(pool_plugin, 'SUPPORTS_FUSED_LRN_CONV', False):
                # ... and the 1x1 convolution behind a blocked MaxPool + LRN rides in that launch, as in an fp32 IR (GoogLeNet: conv1 [blocked] ->
                # pool1 + norm1 -> conv2/3x3_reduce [blocked for conv2/3x3]): pvhip_maxpool3x3_lrn_conv1x1_c8
                tails = {(self._fusion[c_]['relu'] if self._fusion[c_]['relu'] is not None else self._fusion[c_]['add']) for c_ in self._c8_out}
                for pid, lid in self._lrn_pool.items():
                    if G.nodes[pid]['type'] != 'MaxPool' or pid in self._stem_conv:
                        continue
                    src = next((p_ for p_ in G.pred[pid] if G.edges[(p_, pid)]['connection'][3] == 0), None)
                    readers = list(G.successors(lid))
                    if src not in tails or len(readers) != 1 or G.edges[(lid, readers[0])]['connection'][3] != 0:
                        continue
                    cid = readers[0]
                    f = self._fusion.get(cid)
                    if f is None or cid not in self._c8_out or cid in self._siblings or cid in self._fused_away or cid in self._pool_conv \
                            or (f['act'] is not None and f['act'][0] != 'relu'):
                        continue
                    if pool_plugin.lrn_conv_fusable(G.nodes[pid], G.nodes[lid], G.nodes[cid], True):
                        self._stem_conv[pid] = cid
                        self._fused_away.update(n_ for n_ in (cid, f['add'], f['relu']) if n_ is not None)
            if device.conv_f16_c8 == 2 and getattr(conv_plugin, 'SUPPORTS_C8_MODULES', False):
                self.plan_c8_modules(conv_plugin)
        self.order_for_locality()

    def plan_c8_modules(self, conv_plugin):
        """FP16 IRs, second step (default; PVHIP_CONV_F16_C8=1: only the tensors between a 1x1 convolution and the 3x3 / 5x5 behind it): whole inception modules on blocked fp16 tensors.  A channel Concat whose members are
        fused convolution chains that all (a) read a tensor that WILL be blocked -- the previous module's blocked Concat, a 3x3 MaxPool of
        one, a 3x3_reduce / 5x5_reduce tensor (`_c8_out`), or the converted entry tensor -- and (b) run on pvhip_conv2d_f16_c8_multi, gets a
        blocked buffer (`_c8_concat`).  The tensor the first module reads is converted once (`_c8_entry`).  A reader that does not take
        the blocked layout densifies by itself (device.as_device): only writers need this plan."""
        G = self.ienet.G
        pool_plugin, lrn_plugin = (self.ienet.ie.plugins.plugins.get(t) for t in ('MaxPool', 'LRN'))
        pool_plugin = pool_plugin if hasattr(pool_plugin, 'blocked_ok') else None
        lrn_plugin = lrn_plugin if hasattr(lrn_plugin, 'blocked_ok') else None
        src_of = lambda nid: next((p_ for p_ in G.pred[nid] if G.edges[(p_, nid)]['connection'][3] == 0), None)   # noqa: E731
        blocked = set()
        for cid in self._c8_out:
            f = self._fusion[cid]
            blocked.add(f['relu'] if f['relu'] is not None else f['add'])

        def data_src(cid):
            pooled = self._pool_conv.get(cid)
            return src_of(pooled[0]) if pooled is not None else src_of(cid)

        def member_ok(cid, assume=None):
            f = self._fusion.get(cid)
            if f is None or cid in self._pre_add or (f['act'] is not None and f['act'][0] != 'relu'):
                return False
            pooled = self._pool_conv.get(cid)
            src = data_src(cid)
            if src is None or not (src in blocked or src == assume):
                return False
            return conv_plugin.c8_module_member_ok(G.nodes[cid], G.nodes[pooled[0]] if pooled is not None else None)

        members_of = {}
        for cid, f in self._fusion.items():
            if f['into'] is not None:
                members_of.setdefault(f['into'][0], []).append(cid)
        for nid in self.list_schedule:
            node = G.nodes[nid]
            if node['type'] == 'MaxPool':
                src = src_of(nid)
                folded = G.nodes[self._lrn_pool[nid]] if nid in self._lrn_pool else None
                # MaxPool.blocked_ok / LRN.blocked_ok: the predicates the plugins themselves decide with at run time
                if src in blocked and nid not in self._lrn_pool.values() and pool_plugin is not None and pool_plugin.blocked_ok(node, folded):
                    blocked.add(nid)          # the plugin pools a blocked tensor as it is (a folded pool hands its input on)
                    if nid in self._lrn_pool:
                        blocked.add(self._lrn_pool[nid])      # MaxPool + LRN on the blocked tensor: the folded LRN carries it
            elif node['type'] == 'LRN' and nid in self._lrn_pool and src_of(nid) in blocked and lrn_plugin is not None \
                    and lrn_plugin.blocked_ok(node, G.nodes[self._lrn_pool[nid]]):
                blocked.add(self._lrn_pool[nid])              # LRN + MaxPool on a blocked tensor: the folded MaxPool carries it
            elif node['type'] == 'Concat' and nid in self._concat_direct:
                members = members_of.get(nid, [])
                if not members or int(next(iter(node['output'].values()))['dims'][1]) % 16 != 0:
                    continue          # (a blocked tensor holds whole 16-channel stages; its members write whole 8-channel blocks: c8_module_member_ok)
                # the first module: the tensor its 1x1 arms read is not blocked yet -- it is converted if that makes the module blocked
                entry = None
                srcs = {data_src(m) for m in members}
                outside = [s_ for s_ in srcs if s_ not in blocked]
                if len(outside) == 1 and G.nodes[outside[0]]['type'] not in ('Convolution', 'Concat', 'Const', 'Parameter'):
                    entry = outside[0]
                    readers = list(G.successors(entry))
                    folded_pools = {p[0] for p in self._pool_conv.values()}
                    if not all(r in self._fusion or r in folded_pools for r in readers):
                        entry = None
                if all(member_ok(m, assume=entry) for m in members):
                    self._c8_concat.add(nid)
                    blocked.add(nid)
                    if entry is not None:
                        self._c8_entry.add(entry)
                        blocked.add(entry)

    def order_for_locality(self):
        """Another legal order of the same list schedule (round 4; scripts/exp_hoist.py).  The reference's sweep (:218-242) runs the
        arms of an inception module as 1x1 / 3x3_reduce / 5x5_reduce (here: ONE sibling launch), 3x3, 5x5, pool -> pool_proj.  The
        arms behind a sibling launch are mutually independent, so:
          * MaxPool + pool_proj, which reads the SAME module input as the sibling launch (38-205 MB), goes right behind it -- the
            tensor is then still in L2 / the 256 MB Infinity Cache instead of behind the traffic of the 3x3 and 5x5 arms
            (its seven launches 0.492 -> 0.444 ms);
          * the remaining arms run in ascending order of their output (5x5 before 3x3): the largest part of the module's output is
            written last, closest to the next module's reads (all convolutions 4.98 -> 4.88-4.91 ms, one infer() -2 %).
        Same launches, same tensors, same bits; PVHIP_SCHEDULE_LOCALITY=0 keeps the reference's order."""
        if os.environ.get('PVHIP_SCHEDULE_LOCALITY', '1') == '0' or not self._siblings:
            return
        G = self.ienet.G
        order = list(self.task_list)
        position = {t: i for i, t in enumerate(order)}
        src_of = lambda cid: next((p_ for p_ in G.pred[cid] if G.edges[(p_, cid)]['connection'][3] == 0), None)   # noqa: E731
        for lead in sorted(self._siblings, key=position.get):
            members = [lead] + list(self._siblings[lead])
            tails = set()
            for m in members:
                f = self._fusion[m]
                tails.add(f['relu'] if f['relu'] is not None else f['add'])
            arms = [t for t in order if t not in self._fused_away and G.nodes[t]['type'] == 'Convolution' and src_of(t) in tails
                    and t in self._fusion and t not in self._siblings]
            pooled = [c for c, (_, psrc) in self._pool_conv.items() if psrc == src_of(lead)]
            out_elems = lambda t: int(np.prod(next(iter(G.nodes[t]['output'].values()))['dims']))                # noqa: E731
            arms.sort(key=lambda t: (out_elems(t), position[t]))
            moved = pooled + arms
            if not moved:
                continue
            # every moved task depends on the lead's launch (or on the lead's own input) and on constants only: any order behind the lead is legal
            legal = all(all(G.nodes[p_]['type'] == 'Const' or p_ in tails or (t in pooled and p_ == self._pool_conv[t][0]) for p_ in G.pred[t]) for t in moved)
            if not legal:
                continue
            # a moved unit = the convolution with the nodes folded into it (its MaxPool in front, its Add / ReLU behind), so that the
            # list stays a topological order of the WHOLE graph
            units = []
            for t in moved:
                f = self._fusion[t]
                units += ([self._pool_conv[t][0]] if t in pooled else []) + [t] + [n_ for n_ in (f['add'], f['relu']) if n_ is not None]
            gone = set(units)
            rest = [t for t in order if t not in gone]
            lead_chain = [lead] + [n_ for m in members for n_ in ((m,) if m != lead else ()) + (self._fusion[m]['add'], self._fusion[m]['relu']) if n_ is not None]
            at = max(rest.index(n_) for n_ in lead_chain) + 1
            order = rest[:at] + units + rest[at:]
        self.task_list = order

    def prepare_inputs_for_task(self, task) -> dict:
        """{sink port: tensor} gathered from the predecessors' output ports, in edge order."""
        G = self.ienet.G
        inputs = {}
        for pred in G.pred[task]:
            src_node, src_port, _, sink_port = G.edges[(pred, task)]['connection']
            inputs[sink_port] = G.nodes[src_node]['output'][src_port]['data']
        return inputs

    def plan_streams(self):
        """Static stream assignment for the current task list (the reference's list scheduler
        runs the branches of a module one after the other, :259-292).  The consumers of a tensor are ranked by
        the estimated time of the arm each one starts (the chain of single-consumer nodes behind it); the
        heaviest stays on the stream the tensor was produced on, the others go to the next streams, so the arms
        of a fan-out run side by side.  A tensor assembled by several producers (an eliminated Concat) counts as
        produced on the stream of the producer expected to finish last.  Both rules keep the critical path of
        consecutive modules on ONE stream: its kernels follow each other without waiting for a cross-stream event
        (measured 25-40 us per join), which only the lighter arms pay.  Returns (stream of task, tasks to wait
        for, tasks that must record an event) or None when not applicable."""
        registry = self.ienet.ie.plugins.plugins
        n = max(1, min(int(self.compute_streams), 8 - self.stream_base))
        if (n <= 1 and self.stream_base == 0 and not self.defer_sync) or not all(getattr(sys.modules.get(m.__package__), 'DEVICE_STREAMS', False) for m in registry.values()):
            return None                  # some plugin of the set computes on the host
        key = (tuple(self.task_list), frozenset(self._fused_away), n)
        plan = self._stream_plans.get(key)
        if plan is not None:
            return plan
        G = self.ienet.G
        owner = {}                       # folded-away node -> the dispatched node that writes its tensor
        for cid, f in self._fusion.items():
            for nid in (f['add'], f['relu']):
                if nid is not None:
                    owner[nid] = cid
        for lid, pid in self._lrn_pool.items():
            owner[pid] = lid
        for lead, sibs in self._siblings.items():
            for sid in sibs:
                for nid in (sid, self._fusion[sid]['add'], self._fusion[sid]['relu']):
                    if nid is not None:
                        owner[nid] = lead
        for pid, cid in self._stem_conv.items():       # the 1x1 convolution chain behind MaxPool + LRN: written by the MaxPool's launch
            for nid in (cid, self._fusion[cid]['add'], self._fusion[cid]['relu']):
                if nid is not None:
                    owner[nid] = pid

        folded_adds = {pool_id: src_id for pool_id, src_id in self._pool_conv.values()}      # MaxPools folded into their consumer's fetch
        folded_adds.update({add_id: src_id for add_id, _, src_id in self._pre_add.values()})  # Adds folded into a padding pass

        def producers(nid):
            if nid in folded_adds:           # an Add folded into its consumer's fetch: whoever wrote the Add's input
                return producers(folded_adds[nid])
            if nid in self._concat_direct and nid in self._fused_away:
                return [p for pred in G.pred[nid] for p in producers(pred)]
            if nid in owner:
                return [owner[nid]]
            if G.nodes[nid]['type'] in ('Const', 'Parameter'):
                return []                # uploads are synchronous (or the tensor is already resident)
            return [nid]

        def prod(dims):
            out = 1
            for d in dims:
                out *= int(d)
            return out

        def cost(task, alone=False):     # rough device time of a task in microseconds (ranking only)
            if not alone and task in self._siblings:
                return cost(task, True) + sum(cost(s_, True) for s_ in self._siblings[task])
            node = G.nodes[task]
            out = prod(next(iter(node['output'].values()))['dims']) if node.get('output') else 0
            if node['type'] == 'Convolution':
                k = node['input'][1]['dims']
                return 2.0 * out * k[1] * k[2] * k[3] / 100e6 + 4.0 * out / 4.5e6
            if node['type'] == 'MatMul':
                return 2.0 * out * node['input'][0]['dims'][-1] / 20e6
            inp = prod(node['input'][0]['dims']) if node.get('input') else 0
            return 4.0 * (inp + out) / 4.5e6

        dispatched = [t for t in self.task_list
                      if t not in self._fused_away and G.nodes[t]['type'] not in ('Const', 'Parameter')]
        position = {t: i for i, t in enumerate(dispatched)}

        def tail(task):                  # graph node whose output port carries the tensor the task writes
            if task in self._stem_conv:
                f = self._fusion[self._stem_conv[task]]
                return f['relu'] if f['relu'] is not None else (f['add'] if f['add'] is not None else self._stem_conv[task])
            if task in self._lrn_pool:
                return self._lrn_pool[task]
            f = self._fusion.get(task)
            if f is None:
                return task
            return f['relu'] if f['relu'] is not None else f['add']

        def consumers(nid):              # dispatched tasks that read the tensor of graph node nid
            out = []
            for succ in G.successors(nid):
                if (succ in self._concat_direct and succ in self._fused_away) or succ in folded_adds:
                    out += [c_ for c_ in consumers(succ) if c_ not in out]     # (a folded Add / MaxPool hands the tensor on)
                elif succ in position and succ not in out:
                    out.append(succ)
            return sorted(out, key=position.get)

        def joins(task):                 # the task writes into a tensor that other tasks write too
            f = self._fusion.get(task)
            return f is not None and f['into'] is not None

        arm_memo = {}

        def arm_cost(task):              # the task plus the chain of sole consumers behind it, up to the next fork / join
            if task not in arm_memo:
                total, cur = 0.0, task
                while True:
                    total += cost(cur)
                    nxt = consumers(tail(cur))
                    if joins(cur) or len(nxt) != 1:
                        break
                    srcs = {p for pred in G.pred[nxt[0]] for p in producers(pred)}
                    if srcs != {cur}:
                        break
                    cur = nxt[0]
                arm_memo[task] = total
            return arm_memo[task]

        stream_of, waits, records, rank_of, finish, width_of = {}, {}, set(), {}, {}, {}
        for task in dispatched:
            preds = sorted(G.pred[task], key=lambda p: G.edges[(p, task)]['connection'][3])
            primary = next((p for p in preds if producers(p)), None)
            while primary in folded_adds:        # read through a folded Add / MaxPool: the arms fork at ITS input
                primary = folded_adds[primary]
            if primary is None:
                stream_of[task] = 0
                finish[task] = cost(task)
            else:
                if primary not in rank_of:           # heaviest arm first; schedule order breaks ties
                    cons, writers_ = consumers(primary), producers(primary)
                    if len(writers_) == 1 and writers_[0] in self._siblings:
                        # one launch wrote several tensors: the arms behind ALL of them fan out from its stream
                        cons = []
                        for t in [writers_[0]] + list(self._siblings[writers_[0]]):
                            if not joins(t):         # (a tensor assembled with others is ranked when its last writer is known)
                                cons += [c for c in consumers(tail(t)) if c not in cons]
                    arms = sorted(cons, key=lambda t: (-arm_cost(t), position[t]))
                    # the lighter arms behind a sibling launch skip the streams taken by the arms the launch itself forked with
                    skip = width_of.get(writers_[0], 1) - 1 if (len(writers_) == 1 and writers_[0] in self._siblings) else 0
                    rank_of[primary] = {t: (j + skip if j else 0) for j, t in enumerate(arms)}
                srcs = producers(primary)
                base = max(srcs, key=lambda p: (finish[p], -position[p]))     # the producer expected to finish last
                stream_of[task] = (stream_of[base] + rank_of[primary].get(task, 0)) % n
                width_of[task] = len(rank_of[primary])
                finish[task] = finish[base] + cost(task)
            deps = []
            for pred in preds:
                for p in producers(pred):
                    if stream_of[p] != stream_of[task] and p not in deps:
                        deps.append(p)
            waits[task] = deps
            records.update(deps)
        plan = (stream_of, waits, records)
        self._stream_plans[key] = plan
        return plan

    def recorded_waits(self):
        """The cross-stream waits a RECORDING of the current plan makes, in dispatch order, as (how, waiting stream, event's stream,
        producer task) with how = 'plain' | 'relay' (CaptureStreamModel) -- what _dispatch_tasks issues while a capture is open,
        computed from the plan alone (no device)."""
        plan = self.plan_streams()
        if plan is None:
            return [], CaptureStreamModel()
        stream_of, waits, _ = plan
        model, out = CaptureStreamModel(), []
        for task in self.task_list:
            if task in self._fused_away or task not in stream_of:
                continue
            for dep in waits[task]:
                out.append((model.wait(stream_of[task], stream_of[dep]), stream_of[task], stream_of[dep], dep))
        return out, model

    def recording_rings(self) -> bool:
        """True when a recording of the current plan would leave a ring in the runtime's parallel-stream lists."""
        return self.recorded_waits()[1].has_ring()

    def run_tasks(self, verbose: bool = False):
        G = self.ienet.G
        registry = self.ienet.ie.plugins.plugins
        times = []
        open_run = None
        self._infer_serial += 1
        self._recycle_events()
        plan = self.plan_streams()
        if plan is not None:
            from . import device
            stream_of, waits, records = plan
            done_events, spare, current = {}, self.__dict__.setdefault('_order_events', []), 0
            held = self.__dict__.setdefault('_events_in_flight', [])
            spare.extend(held)           # the pass that recorded them has been waited for by now
            del held[:]
            base = self.stream_base
            # blocks allocated during this pass and freed before it has finished on the device (workspaces) are
            # parked until it has; the previous pass's outputs, replaced as we go, are reusable at once
            epoch = self._open_epoch = device.pool_epoch_begin()
            device.select_stream(base)
        try:
            self._dispatch_tasks(G, registry, plan, times, verbose)
        except BaseException:
            # a plugin raised in the middle of a pass: leave the device in a defined state -- every stream drained, stream 0
            # current, the allocation epoch closed (its parked blocks back in the pool) -- and let the error travel on
            if plan is not None:
                from . import device
                try:
                    device.select_stream(0)
                    device.synchronize()
                    device.pool_epoch_dispatched()
                    device.pool_epoch_end(self._open_epoch)
                except Exception:
                    pass
            raise
        self.last_node_times = times

    def _dispatch_tasks(self, G, registry, plan, times, verbose):
        open_run = None
        if plan is not None:
            from . import device
            stream_of, waits, records = plan
            done_events, spare, current = {}, self.__dict__.setdefault('_order_events', []), 0
            held = self.__dict__.setdefault('_events_in_flight', [])
            base = self.stream_base
            epoch = self._open_epoch
            cap_model = CaptureStreamModel() if self.__dict__.get('_recording') else None
            ops = self.__dict__.get('_stream_ops')      # tests: the cross-stream waits of the pass as they are issued
        for task in self.task_list:
            if task in self._fused_away:
                continue
            node = G.nodes[task]
            node_type = node['type']
            if plan is not None and task in stream_of:
                if stream_of[task] != current:
                    current = stream_of[task]
                    device.select_stream(base + current)
                for dep in waits[task]:
                    how = cap_model.wait(current, stream_of[dep]) if cap_model is not None else 'plain'
                    if ops is not None:
                        ops.append((how, current, stream_of[dep], dep))
                    if how == 'relay':                   # (a recording only: see CaptureStreamModel)
                        device.select_stream(base)
                        done_events[dep].wait()
                        relay = (spare.pop() if spare else device.Event(timed=False)).record()
                        device.select_stream(base + current)
                        relay.wait()
                        held.append(relay)
                    else:
                        done_events[dep].wait()
            pooled_in = self._pool_conv.get(task)
            if pooled_in is not None:        # the folded MaxPool hands its own input on: the kernel pools while it builds its tile
                pool_id, _ = pooled_in
                edge = next(G.edges[(p_, pool_id)]['connection'] for p_ in G.pred[pool_id] if G.edges[(p_, pool_id)]['connection'][3] == 0)
                out = G.nodes[pool_id]['output']
                out[next(iter(out))]['data'] = G.nodes[edge[0]]['output'][edge[1]]['data']
                node['_fuse_pool_in'] = G.nodes[pool_id]
            else:
                node.pop('_fuse_pool_in', None)
            pre_add = self._pre_add.get(task)
            if pre_add is not None:          # the folded Add hands its data input on; its constant rides in the padding pass
                add_id, const_id, src_id = pre_add
                edge = G.edges[(src_id, add_id)]['connection']
                out = G.nodes[add_id]['output']
                out[next(iter(out))]['data'] = G.nodes[edge[0]]['output'][edge[1]]['data']
                node['_pre_add'] = G.nodes[const_id]['output'][0]['data']
            else:
                node.pop('_pre_add', None)
            inputs = self.prepare_inputs_for_task(task) if 'input' in node else {}
            if node_type in ('Convolution', 'MatMul'):
                node['_f16_mfma'] = bool(getattr(self.ienet, 'f16_mfma', False))
            fusion = self._fusion.get(task)
            node.pop('_out_into', None)
            if fusion is not None:
                node['_fuse_bias'] = G.nodes[fusion['bias']]['output'][0]['data']
                node['_fuse_act'] = fusion['act']
                if fusion['into'] is not None:
                    node['_out_into'] = (self._concat_buffer(fusion['into'][0]), fusion['into'][1])
            else:
                node.pop('_fuse_bias', None)
                node.pop('_fuse_act', None)
            sibs = self._siblings.get(task)
            if task in self._c8_out:
                node['_out_c8'] = True
            else:
                node.pop('_out_c8', None)
            if sibs:
                node['_siblings'] = []
                for sid in sibs:
                    sf = self._fusion[sid]
                    node['_siblings'].append({'node': G.nodes[sid], 'inputs': self.prepare_inputs_for_task(sid),
                                              'bias': G.nodes[sf['bias']]['output'][0]['data'],
                                              'into': (self._concat_buffer(sf['into'][0]), sf['into'][1]) if sf['into'] is not None else None,
                                              'c8': sid in self._c8_out})
            else:
                node.pop('_siblings', None)
            pooled = self._lrn_pool.get(task)        # the node folded into this one: a MaxPool behind an LRN, or an LRN behind a MaxPool
            fuse_key = '_fuse_pool' if node_type == 'LRN' else '_fuse_lrn'
            if pooled is not None:
                node[fuse_key] = G.nodes[pooled]
            else:
                node.pop(fuse_key, None)
            stem_conv = self._stem_conv.get(task)        # the 1x1 convolution behind MaxPool + LRN, in the same launch
            if stem_conv is not None:
                sf = self._fusion[stem_conv]
                wsrc = next(G.edges[(p_, stem_conv)]['connection'] for p_ in G.pred[stem_conv] if G.edges[(p_, stem_conv)]['connection'][3] == 1)
                node['_fuse_conv'] = {'node': G.nodes[stem_conv], 'w': G.nodes[wsrc[0]]['output'][wsrc[1]]['data'],
                                      'bias': G.nodes[sf['bias']]['output'][0]['data'], 'act': sf['act'],
                                      'c8': bool(getattr(self.ienet, 'f16_mfma', False))}       # FP16 IRs: on blocked fp16 tensors
            else:
                node.pop('_fuse_conv', None)
            plugin = registry.get(node_type)
            if plugin is None:
                print("ERROR: Operation '{}' (node={}) is not supported.".format(node_type, node['name']))
                sys.exit(-1)
            timed = self.device_timing is not None and (self.device_timing == 'all' or node_type in self.device_timing)
            if timed:
                if open_run is None:
                    open_run = [task, node_type, node['name'], self._event().record(), 0]
                open_run[4] += 1
            elif open_run is not None and node_type not in self.NO_LAUNCH_TYPES:
                open_run = self._close_run(open_run)
            if task in self.pickle_node_args:
                self.dump_node_args(task, node, inputs)
            t0 = time.time()
            res = plugin.compute(node, inputs, kernel_type=self.kernel_type, debug=False)
            dt = time.time() - t0
            if timed and not self.device_timing_runs:
                open_run = self._close_run(open_run)
            if plan is not None and task in records:
                done_events[task] = (spare.pop() if spare else device.Event(timed=False)).record()
            times.append((task, node_type, node['name'], dt))
            if verbose:
                print('{}, {}, {}, {}'.format(task, node_type, node['name'], dt))
            if self.expected_result is not None and node['name'] in self.expected_result and len(res) > 0:
                # the reference's hook (inference_engine.py:284-287 -> common_def.py:71-105); entries in its format
                # {name: [precision, dims, ndarray]} or bare arrays; `expected_rtol` = its rtol of 1 unless the caller tightens it
                common_def.compare_results(node['name'], next(iter(res.values())), self.expected_result, disp_results=False,
                                           rtol=self.expected_rtol)
            if len(res) > 0:
                if self._c8_entry and (task in self._c8_entry or (pooled is not None and pooled in self._c8_entry)):
                    # the tensor the first blocked module reads: converted once, every reader gets the blocked form
                    from . import device as dev_
                    res = {port_id: (dev_.BlockedHalf.from_dense(data) if isinstance(data, dev_.DeviceTensor) and data.ndim == 4 else data)
                           for port_id, data in res.items()}
                for port_id, data in res.items():
                    node['output'][port_id]['data'] = data
                if fusion is not None:
                    fused = next(iter(res.values()))
                    for nid in (fusion['add'], fusion['relu']):
                        if nid is not None:
                            out = G.nodes[nid]['output']
                            out[next(iter(out))]['data'] = fused
                if sibs:                         # the launch wrote the siblings' tensors too
                    for sid, tensor in zip(sibs, node.pop('_sibling_out')):
                        chain = [sid, self._fusion[sid]['add'], self._fusion[sid]['relu']]
                        for nid in chain:
                            if nid is not None:
                                out = G.nodes[nid]['output']
                                out[next(iter(out))]['data'] = tensor
                if pooled is not None:           # the folded node's port carries the tensor
                    out = G.nodes[pooled]['output']
                    out[next(iter(out))]['data'] = next(iter(res.values()))
                if stem_conv is not None:        # ... and so do the ports of the folded convolution chain (what the launch returned IS its output;
                    sf = self._fusion[stem_conv]  # the pooled and the normalised tensor do not exist: their ports hold it only as a placeholder)
                    for nid in (stem_conv, sf['add'], sf['relu']):
                        if nid is not None:
                            out = G.nodes[nid]['output']
                            out[next(iter(out))]['data'] = next(iter(res.values()))
        if open_run is not None:
            self._close_run(open_run)
        if plan is not None:
            # join: stream 0 continues after everything the other streams were given, then the host waits
            # (infer() has read the Result back by now, so this costs nothing) and freed blocks become reusable
            joins = []
            for st in sorted(set(stream_of.values()) - {0}):
                device.select_stream(base + st)
                joins.append((spare.pop() if spare else device.Event(timed=False)).record())
            device.select_stream(base)
            for ev in joins:
                ev.wait()
            held.extend(joins)
            held.extend(done_events.values())
            device.pool_epoch_dispatched()
            if self.defer_sync:          # asynchronous request: wait_done() ends the pass
                self._pending = (epoch, (spare.pop() if spare else device.Event(timed=False)).record())
            else:
                device.select_stream(0)
                device.synchronize()
                device.pool_epoch_end(epoch)

    # ---- hipGraph replay of a whole pass (what the reference's run_tasks loop, :259-292, becomes: one launch call)
    def capture_graph(self, inputs: dict, warm: int = 2, streams=1):
        """Record one forward pass into a hipGraph, for inputs that are resident on the device.  `infer_graph()` then replays it with
        ONE call instead of ~100 dispatches.  The graph holds the addresses of every tensor of the pass: they are kept alive with it
        (`release_graph`).  `streams` = 1 (default): the pass is recorded on one compute stream, a linear chain of launches -- with the
        persistent-grid kernels of this build that replays as fast as the forked form (googlenet-v1 batch 256: 44.5 k images/s either
        way, scripts/time_replay.py).  `streams='plan'`: recorded as the stream plan forks it (`compute_streams`).  Until round 4 that
        killed the process for some plans (stack overflow inside hipStreamEndCapture of ROCm 7.2: its walk over the per-stream lists of
        parallel capture streams meets a ring when non-origin streams wait for each other in both directions over time,
        profiles/r04_capture.md); the dispatcher now relays the ring-closing waits through the origin stream (CaptureStreamModel)."""
        if streams != 'plan':
            saved_streams = self.compute_streams
            self.compute_streams = max(1, int(streams))
            try:
                return self.capture_graph(inputs, warm=warm, streams='plan')
            finally:
                self.compute_streams = saved_streams
        from . import device
        G = self.ienet.G
        if not all(isinstance(v, device.DeviceTensor) for v in inputs.values()):
            raise ValueError('capture_graph needs device-resident inputs (DeviceTensor): their addresses go into the graph')
        self.release_graph()
        for _ in range(max(1, warm)):           # the pool learns every block size of the pass: a capture must not hipMalloc
            self._infer_eager(inputs)
        by_name = {G.nodes[n]['name']: n for n in G.nodes}
        for node_name, val in inputs.items():
            G.nodes[by_name[node_name]]['param'] = val
        results = self.ienet.find_node_by_type('Result')
        for nid, _ in results:
            G.nodes[nid]['comm'] = None
            G.nodes[nid]['_async'] = True       # the Result stays on the device: infer_graph() reads it back
        saved_timing, self.device_timing = self.device_timing, None
        if self.recording_rings():              # (cannot happen with the relays in place: refuse before any HIP call, never crash)
            raise device.PvhipError('capture_graph: this stream plan would close a ring in the runtime\'s parallel-stream lists '
                                    '(hipStreamEndCapture of ROCm 7.2 never returns from it); record on one stream')
        device.select_stream(self.stream_base)
        device.call('pvhip_graph_begin_capture')
        handle = ctypes_void_p()
        first_error = None
        try:
            self.defer_sync = self._recording = True
            try:
                self.run_tasks(False)
            finally:
                self.defer_sync = self._recording = False
                self.device_timing = saved_timing
                for nid, _ in results:
                    G.nodes[nid].pop('_async', None)
        except BaseException as exc:            # noqa: BLE001 -- kept: end_capture below may fail too and must not mask it
            first_error = exc
        try:
            device.select_stream(self.stream_base)
            device.call('pvhip_graph_end_capture', byref(handle))       # (also after an error: the capture must be closed)
        except Exception:                       # noqa: BLE001
            if first_error is None:
                raise
        if first_error is not None:
            if handle.value:
                device.call('pvhip_graph_destroy', ctypes_void_p(handle.value))
            raise first_error
        pending = self.__dict__.pop('_pending', None)   # its event belongs to the graph: nothing to wait for, just close the epoch
        if pending is not None:
            device.pool_epoch_end(pending[0])
        keep = [p['data'] for n in G.nodes for p in G.nodes[n].get('output', {}).values() if 'data' in p]
        keep += [G.nodes[nid]['result'] for nid, _ in results]
        self._graph = {'handle': handle.value, 'inputs': dict(inputs), 'keep': keep,
                       'results': {name: G.nodes[nid]['result'] for nid, name in results}}
        self._graph['by_hand'] = not self.__dict__.get('_auto_graph_busy')     # a recording made by hand is never replaced by infer()
        device.select_stream(0)

    def dump_node_args(self, task, node, inputs):
        """The reference's node-replay hook (`pyopenvino/inference_engine.py:216, 275-278`): `(node, inputs)` of a chosen node,
        pickled as `node_args_<id>.pickle`, so that the node can be run on its own (`test_node_sample.py:1-16`; its fixture
        `resources/node_args_6.pickle` was made this way).  What the file holds is what the REFERENCE's plugins can load: device
        tensors are copied to host ndarrays, and the scheduler's private hints (`_fuse_bias`, `_out_into`, device caches: every key
        that starts with an underscore) stay out, so a fused Convolution replays as the plain Convolution it is in the IR."""
        import pickle
        from . import device

        def plain(obj):
            if isinstance(obj, (device.DeviceTensor, device.ChannelSlice)):
                return np.array(obj)
            if isinstance(obj, dict):
                return {k: plain(v) for k, v in obj.items() if not (isinstance(k, str) and (k.startswith('_') or k == 'comm'))}
            if isinstance(obj, (list, tuple)):
                return type(obj)(plain(v) for v in obj)
            return obj

        with open(os.path.join(self.pickle_dir, 'node_args_{}.pickle'.format(task)), 'wb') as f:
            pickle.dump((plain(node), plain(inputs)), file=f)

    def infer_graph(self, inputs: dict = None) -> dict:
        """Replay the captured pass (for new inputs: copied device-to-device into the captured input tensors first) and return
        {Result name: ndarray} like infer()."""
        from . import device
        import ctypes
        g = self.__dict__.get('_graph')
        if g is None:
            raise RuntimeError('no captured graph: call capture_graph(inputs) first')
        device.select_stream(self.stream_base)
        for name, val in (inputs or {}).items():
            dst = g['inputs'][name]
            if val is dst:
                continue
            src = device.as_device(val)
            if src.shape != dst.shape:
                raise ValueError('input {} has shape {}, the graph was captured for {}'.format(name, src.shape, dst.shape))
            device.call('pvhip_memcpy_d2d', ctypes.c_void_p(dst.ptr), ctypes.c_void_p(src.ptr), dst.nbytes)
        device.call('pvhip_graph_launch', ctypes.c_void_p(g['handle']))
        out = {name: (t.numpy() if hasattr(t, 'numpy') and not isinstance(t, np.ndarray) else np.asarray(t)) for name, t in g['results'].items()}
        device.select_stream(0)
        device.synchronize()
        return out

    def release_graph(self):
        g = self.__dict__.pop('_graph', None)
        if '_auto_graph' in self.__dict__:
            self._auto_graph.update(captured=False, seen=0)
        if g is not None:
            from . import device
            import ctypes
            device.call('pvhip_graph_destroy', ctypes.c_void_p(g['handle']))

    def wait_done(self):
        """Host-side wait for a pass dispatched with defer_sync (its streams have been joined on the base stream)."""
        replayed = self.__dict__.pop('_replay_done', None)
        if replayed is not None:
            replayed.synchronize()
            self.__dict__.setdefault('_event_pool', []).append(replayed)
        pending = self.__dict__.pop('_pending', None)
        if pending is not None:
            from . import device
            epoch, done = pending
            done.synchronize()
            device.pool_epoch_end(epoch)
            self.__dict__.setdefault('_order_events', []).append(done)

    def _concat_buffer(self, cat_id):
        """Output tensor of a Concat whose producers write in place; one fresh tensor per infer."""
        from . import device
        node = self.ienet.G.nodes[cat_id]
        port = next(iter(node['output']))
        if node.get('_buf_serial') != self._infer_serial:
            dims = node['output'][port]['dims']
            # FP16 IRs, module form: the Concat's buffer is fp16 blocked by eight channels and every member writes its range of it
            node['output'][port]['data'] = device.BlockedHalf(dims) if cat_id in self._c8_concat else device.DeviceTensor.empty(dims)
            node['_buf_serial'] = self._infer_serial
        return node['output'][port]['data']

    # ---- device-side per-node timing (hipEvents on the compute stream; cf. the time.time() bracket :279-283)
    def _event(self):
        from . import device
        pool = self.__dict__.setdefault('_event_pool', [])
        return pool.pop() if pool else device.Event()

    NO_LAUNCH_TYPES = ('Const', 'Parameter', 'Reshape')   # their compute() puts nothing on the stream

    def _close_run(self, run):
        task, node_type, name, e0, count = run
        self._timed.append((task, node_type, name, e0, self._event().record(), count))
        return None

    def _recycle_events(self):
        pool = self.__dict__.setdefault('_event_pool', [])
        for _, _, _, e0, e1, _ in self._timed:
            pool.extend((e0, e1))
        self._timed = []

    def device_times_ms(self, with_counts: bool = False):
        """[(node id, type, name, milliseconds)] for the brackets of the last run_tasks (synchronises).  With
        `device_timing_runs` a bracket spans a run of consecutive bracketed nodes: id / type / name are those of
        its first node and `with_counts=True` appends the number of nodes it covers."""
        out = []
        for task, node_type, name, e0, e1, count in self._timed:
            e1.synchronize()
            row = (task, node_type, name, e0.elapsed_ms(e1))
            out.append(row + (count,) if with_counts else row)
        return out

    def infer_until(self, inputs: dict, node_names) -> dict:
        """Run only what is needed to produce the outputs of the named nodes (e.g. the SSD backbone up to its
        box / class heads, whose host-side PriorBox / DetectionOutput consumers are out of scope) and return
        {node name: tensor of its first output port} -- device tensors are returned as they are."""
        G = self.ienet.G
        by_name = {G.nodes[n]['name']: n for n in G.nodes}
        targets = [by_name[name] for name in node_names]
        needed = set(targets)
        for t in targets:
            needed.update(nx.ancestors(G, t))
        for node_name, val in inputs.items():
            if node_name in by_name:
                G.nodes[by_name[node_name]]['param'] = val
        full, fa = self.task_list, self._fused_away
        try:
            self.task_list = [t for t in full if t in needed]
            # a fused chain must be wholly inside the sub-graph, or be run unfused
            keep = {}
            for cid, f in self._fusion.items():
                chain = [cid, f['add']] + ([f['relu']] if f['relu'] is not None else [])
                if all(c in needed for c in chain):
                    keep[cid] = dict(f, into=None)      # Concat elimination is not applied to sub-graphs
            saved = (self._fusion, self._fused_away, self._concat_direct, self._lrn_pool, self._siblings, self._pool_conv, self._pre_add)
            self._siblings, self._pool_conv, self._pre_add = {}, {}, {}      # sub-graph runs launch every convolution (MaxPool, Add) on its own
            self._fusion = {c: f for c, f in keep.items()}
            self._fused_away = {n for f in self._fusion.values() for n in (f['add'], f['relu']) if n is not None}
            self._concat_direct = {}
            self._lrn_pool = {l: p_ for l, p_ in self._lrn_pool.items() if l in needed and p_ in needed and l not in targets}
            self._fused_away |= set(self._lrn_pool.values())
            try:
                self.run_tasks(False)
            finally:
                self._fusion, self._fused_away, self._concat_direct, self._lrn_pool, self._siblings, self._pool_conv, self._pre_add = saved
        finally:
            self.task_list = full
        out = {}
        for name, t in zip(node_names, targets):
            ports = G.nodes[t]['output']
            out[name] = ports[next(iter(ports))]['data']
        return out

    # ---- replay instead of dispatch.  A forward pass is ~100 plugin calls = 0.8-0.9 ms of Python + ctypes per pass; with inputs that
    # are resident on the device -- the SAME tensors from call to call -- the pass is the same list of launches on the same addresses
    # every time, so after AUTO_GRAPH_AFTER identical eager passes infer() records it into a hipGraph once (capture_graph) and replays it from then on with one call
    # (infer_graph: bit-identical, tests/test_hip_models.py).  Anything that makes a pass differ -- other input tensors, another stream
    # plan or fusion plan, re-read PVHIP_* settings, hooks that look at single nodes (verbose, expected_result, pickle_node_args,
    # device_timing), a sharded batch -- runs eagerly, and a changed key drops the recording.  PVHIP_AUTO_GRAPH=0 turns it off.
    AUTO_GRAPH_AFTER = 2

    def _auto_graph_key(self, inputs, verbose, gathers_later=False):
        from . import device
        if verbose or os.environ.get('PVHIP_AUTO_GRAPH', '1') == '0' or self.__dict__.get('_auto_graph_busy'):
            return None
        if self.expected_result is not None or self.pickle_node_args or self.device_timing is not None or self.defer_sync:
            return None
        if not gathers_later and self.comm is not None and getattr(self.comm, 'world', 1) > 1:
            return None                             # (a request gathers its shards in wait(), after the recorded pass)
        if not inputs or not all(isinstance(v, device.DeviceTensor) for v in inputs.values()):
            return None
        registry = self.ienet.ie.plugins.plugins
        if not all(getattr(sys.modules.get(m.__package__), 'DEVICE_STREAMS', False) for m in registry.values()):
            return None
        if '_graph_safe' not in self.__dict__:      # every layer type of this network says its compute() can be recorded
            G = self.ienet.G                        # (a foreign plugin set does not: its pass stays eager)
            self._graph_safe = all(getattr(registry.get(G.nodes[n]['type']), 'GRAPH_CAPTURE_SAFE', False) for n in G.nodes)
        if not self._graph_safe:
            return None
        # (the recording reads the inputs where they lie: another tensor is another recording, never a copy into the caller's tensor)
        return (tuple(sorted((k, tuple(v.shape), v.ptr) for k, v in inputs.items())), self.compute_streams, self.stream_base,
                self.__dict__.get('_plan_serial', 0), self.fuse_epilogues, device.settings_serial, self.kernel_type)

    def _graph_for(self, inputs, verbose=False, gathers_later=False):
        """The recording that replays this pass, or None: the pass is dispatched eagerly (and counted; the recording is made on the
        call after AUTO_GRAPH_AFTER identical eager ones)."""
        key = self._auto_graph_key(inputs, verbose, gathers_later)
        state = self.__dict__.setdefault('_auto_graph', {'key': None, 'seen': 0, 'failed': False})
        g = self.__dict__.get('_graph')
        if key is None or state['failed'] or (g is not None and g.get('by_hand')):
            return None
        if state['key'] != key:
            if state['key'] is not None and self.__dict__.get('_graph') is not None and state.get('captured'):
                self.release_graph()
            state.update(key=key, seen=0, captured=False)
        if state.get('captured') and self.__dict__.get('_graph') is not None:
            return self._graph
        state['seen'] += 1
        if state['seen'] <= self.AUTO_GRAPH_AFTER:
            return None
        self._auto_graph_busy = True
        saved_comm = self.comm
        if gathers_later:
            self.comm = None                # (the warm pass in front of the recording must not gather either: wait() does)
        try:
            self.capture_graph(inputs, warm=1)
            state['captured'] = True
        except Exception as exc:           # noqa: BLE001 -- replay is an optimisation: say why it is off, keep computing
            state['failed'] = True
            print('pyopenvino_amd: hipGraph replay of infer() disabled for this network ({}: {})'.format(type(exc).__name__, exc), file=sys.stderr)
            return None
        finally:
            self._auto_graph_busy = False
            self.comm = saved_comm
        return self._graph

    def infer(self, inputs: dict, verbose: bool = False) -> dict:
        if self._graph_for(inputs, verbose) is None:
            return self._infer_eager(inputs, verbose)
        self.last_node_times = []
        return self.infer_graph(inputs)

    def launch_graph(self, g):
        """Asynchronous replay for an infer request: the recorded pass goes to this network's stream with one call; `wait_done()`
        waits for the event behind it."""
        from . import device
        import ctypes
        device.select_stream(self.stream_base)
        device.call('pvhip_graph_launch', ctypes.c_void_p(g['handle']))
        self._replay_done = self._event().record()
        device.select_stream(0)
        self.last_node_times = []

    def _infer_eager(self, inputs: dict, verbose: bool = False) -> dict:
        G = self.ienet.G
        by_name = {G.nodes[n]['name']: n for n in G.nodes}
        for node_name, val in inputs.items():
            if node_name in by_name:
                G.nodes[by_name[node_name]]['param'] = val
        for nid, _ in self.ienet.find_node_by_type('Result'):
            G.nodes[nid]['comm'] = self.comm
        if verbose:
            print('# node_id node_name time (sec)')
        t0 = time.time()
        self.run_tasks(verbose)
        if verbose:
            print('@TOTAL_TIME,', time.time() - t0)
        return {name: G.nodes[nid]['result'] for nid, name in self.ienet.find_node_by_type('Result')}
