"""Calibration statistics over the calibration set — the GPU-resident replacement of
dipoorlet/forward_net.py:192-342 (forward_get_minmax / forward_get_hist / forward_net_octav) and its
.bin loader (:459-464).

Reference flow per image: ORT forward with every node output exposed -> copy all activations to the
host -> numpy reductions appended to Python lists.  Here: B images per forward, activations stay in
HBM, one batched HIP launch per statistic over the whole tensor set, persistent device accumulators.
The functions keep the reference's names, arguments and return shapes (dict keyed by tensor name) so
tensor_cali and tests read the same; where the reference returns one entry per image and only their
min / max / sum is ever consumed (ranges, histograms), the list holds the already-reduced value unless
`per_image=True` asks for the full lists.

`onnx_graph` is anything offering the reference ONNXGraph's calibration-facing surface:
    .network_inputs            list of input names
    .get_tensor_shape(name)    model input shape (batch dim 1)
    .make_session(args)        -> ActivationSession  (replaces ort.InferenceSession(...all outputs...))
"""
import numpy as np
import torch

from . import ops
from .dist_helper import shard_range
from .platform_settings import platform_setting_table
from .utils import logger, mark

DEFAULT_BATCH = 64   # images per forward (ResNet-50's fp32 forward runs 12 % more images per second than at 32; the statistics kernels run at the
                     # same fraction of the HBM roofline at 32 and 64: same-box A/B of the CLI, scripts/e2e_dbg.sh)


class ActivationSession:
    """What a graph executor must provide: one forward of a batch -> every calibration tensor.

    tensor_names     network inputs first, then every node output in graph order (forward_net.py:220-235)
    elems_per_image  elements of each tensor for ONE image
    run(inputs)      inputs: {name: device tensor [B, ...]} -> list of contiguous fp32 device tensors
                     [B, ...] aligned with tensor_names (the inputs themselves included)
    """
    tensor_names = ()
    elems_per_image = ()

    def run(self, inputs):
        raise NotImplementedError


def auto_batch(elems_per_image, target_bytes=8e9, largest=DEFAULT_BATCH):
    """Calibration images per forward when --calib_batch is not given: about 8 GB of exposed activations per batch, a power of
    two, at most 64 — ResNet-50 (106 MB per image): 64, where its fp32 forward runs 12 % more images per second than at 32;
    ViT-B/16 (532 MB per image): 16 (measured through the CLI, N = 256: 32 -> 707 images/s, 16 -> 668, 64 -> 606 with 34 GB
    per batch and three batches in flight).  The statistics kernels run at the same fraction of the roofline from 8 images up."""
    per_image = 4.0 * max(1, sum(int(e) for e in elems_per_image))
    want = max(1.0, target_bytes / per_image)
    b = 1
    while b * 2 <= largest and b * 2 <= want * 1.42:      # (to the nearer power of two on a log scale)
        b *= 2
    return b


def input_data_generator(input_dir, input_name_list, data_st_idx, data_ed_idx):
    """forward_net.py:459-464 — one dict {input_name: flat fp32 array} per calibration image, read from
    `{input_dir}/{input_name}/{idx}.bin` (raw little-endian fp32)."""
    for idx in range(data_st_idx, data_ed_idx):
        yield {n: np.fromfile(f"{input_dir}/{n}/{idx}.bin", "float32") for n in input_name_list}


def stage_input_batch(input_dir, input_names, shapes, idx0, idx1, pinned):
    """HOST: images [idx0, idx1) of every network input read into one (pinned) staging tensor per input.
    Returns {name: (host tensor [b, per_image], device shape)}."""
    out = {}
    b = idx1 - idx0
    for n in input_names:
        shape = tuple(int(d) for d in shapes[n])
        per = int(np.prod(shape))
        stage = torch.empty((b, per), dtype=torch.float32, pin_memory=pinned)
        sv = stage.numpy()
        for j, idx in enumerate(range(idx0, idx1)):
            a = np.fromfile(f"{input_dir}/{n}/{idx}.bin", "float32")
            if a.size != per:
                raise ValueError(f"{input_dir}/{n}/{idx}.bin holds {a.size} fp32 values, model input needs {per}")
            sv[j] = a
        lead = shape[0] if len(shape) > 0 else 1
        full = (b * lead,) + shape[1:] if len(shape) > 1 else (b * per,)
        out[n] = (stage, full)
    return out


def load_input_batch(input_dir, input_names, shapes, idx0, idx1, device):
    """Images [idx0, idx1) of every network input as device tensors [B, *shape[1:]]: files are read
    into one pinned staging buffer per input and copied with a single async H2D transfer."""
    staged = stage_input_batch(input_dir, input_names, shapes, idx0, idx1, device.type == "cuda")
    return {n: h.to(device, non_blocking=True).reshape(full) for n, (h, full) in staged.items()}


WALL = {}    # host wall seconds of the calibration phase's parts (reported by --timing_json): where a run's time goes outside the GPU


class wall:
    """with wall("name"): ... — adds the block's host wall seconds to WALL[name]."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        import time
        self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        import time
        WALL[self.name] = WALL.get(self.name, 0.0) + time.perf_counter() - self.t0


class CalibrationRun:
    """One rank's sweep(s) over its shard of the calibration set."""
    last = None    # the most recent run of this process WHEN --timing_json asks for it (its timing() is what gets reported;
                   # __main__ clears it once written: a run pins its session, accumulators and every plan's scratch)

    def __init__(self, onnx_graph, args, session=None):
        if getattr(args, "timing_json", None):
            CalibrationRun.last = self
        self.graph = onnx_graph
        self.args = args
        self.batch = int(getattr(args, "calib_batch", None) or 0)      # 0: chosen from the graph's size, once the session knows it
        self.st, self.ed = shard_range(args.data_num, args.rank, args.world_size)
        self.ingest_s = 0.0
        # the .bin files of the first batches are read (pinned staging) WHILE the session is built: the reader needs the graph's
        # declared input shapes only (and the batch size: with an automatic one it starts right behind the session)
        self._reader = self._start_reader(torch.cuda.is_available()) if self.batch else None
        mark("run:reader_started")
        if session is None:
            def first_batch(sess):     # images of this run's first forward: the session warms the libraries up for that, beside its build
                b = self.batch or auto_batch(sess.elems_per_image)
                return max(0, min(b, self.ed - self.st))
            import inspect      # (a graph object of the caller's own may offer the reference's make_session(args) only)
            kw = {"first_batch": first_batch} if "first_batch" in inspect.signature(onnx_graph.make_session).parameters else {}
            with wall("session_build_s"):      # shapes (host rules), node schedule, weights to the device (one transfer)
                session = onnx_graph.make_session(args, **kw)
        mark("run:session_built")
        if not self.batch:
            self.batch = auto_batch(session.elems_per_image)
            self._reader = self._start_reader(torch.cuda.is_available())
        self.session = session
        if self.ed > self.st and hasattr(session, "prewarm_convs"):
            session.prewarm_convs(min(self.batch, self.ed - self.st))     # (helper threads; returns at once)
        self.names = list(self.session.tensor_names)
        self.elems = [int(e) for e in self.session.elems_per_image]
        self.T = len(self.names)
        self.device = torch.device("cuda", torch.cuda.current_device())
        self._plans = {}
        budget_gb = float(getattr(args, "resident_gb", 160.0))
        self._budget = int(budget_gb * 2**30)
        self._resident = []  # tensor sets kept in HBM between pass 1 and pass 2
        self._resident_ok = True
        self._resident_bytes = 0
        # where a run's time goes (--timing_json): host seconds reading .bin files (ingest_s), GPU milliseconds (HIP events on
        # the launch stream) of the network forward and of the statistics kernels
        self._events = {"forward": [], "statistics": []}

    def timed(self, phase):
        """Context manager: HIP events around the GPU work launched inside, summed by timing()."""
        run = self

        class _T:
            def __enter__(self):
                self.e0 = torch.cuda.Event(enable_timing=True)
                self.e1 = torch.cuda.Event(enable_timing=True)
                self.e0.record()

            def __exit__(self, *exc):
                self.e1.record()
                run._events[phase].append((self.e0, self.e1))
        return _T()

    def timing(self):
        """HOST (synchronises): {'ingest_s', 'forward_gpu_s', 'statistics_gpu_s', 'images'} of this rank so far."""
        torch.cuda.synchronize()
        out = {"images": self.n_images(), "ingest_host_s": self.ingest_s, "host_wall": {k: round(v, 4) for k, v in WALL.items()}}
        for k, evs in self._events.items():
            out[k + "_gpu_s"] = sum(a.elapsed_time(b) for a, b in evs) * 1e-3
        fw = [a.elapsed_time(b) * 1e-3 for a, b in self._events["forward"]]
        out["forward_batches_ms"] = [round(1e3 * x, 2) for x in fw]
        ms = torch.cuda.memory_stats(self.device)
        free_b, total_b = torch.cuda.mem_get_info(self.device)
        out["allocator"] = {"device_allocs": ms.get("num_device_alloc"), "device_frees": ms.get("num_device_free"),
                            "alloc_retries": ms.get("num_alloc_retries"), "reserved_peak_gb": round(ms.get("reserved_bytes.all.peak", 0) / 1e9, 2),
                            "device_free_gb": round(free_b / 1e9, 1), "device_total_gb": round(total_b / 1e9, 1)}
        if len(fw) > 2:   # the first batch carries the one-time costs (MIOpen kernel loading / algorithm choice)
            out["forward_first_batch_gpu_s"] = fw[0]
            out["forward_steady_images_per_s"] = self.batch * (len(fw) - 1) / max(sum(fw[1:]), 1e-9)
        return out

    def plan(self, b):
        p = self._plans.get(b)
        if p is None:
            p = self._plans[b] = ops.TensorSetPlan(self.elems, b, self.device)
        return p

    def n_images(self):
        return self.ed - self.st

    def batches(self):
        i = self.st
        while i < self.ed:
            j = min(i + self.batch, self.ed)
            yield i, j
            i = j

    def _start_reader(self, pinned):
        """A helper thread reads the shard's .bin files batch by batch into (pinned) staging memory, at most two batches ahead
        (file I/O releases the GIL).  Returns (queue, thread, n_batches) or None for an empty shard."""
        import queue
        import threading
        shapes = {n: self.graph.get_tensor_shape(n) for n in self.graph.network_inputs}
        bounds = list(self.batches())
        if not bounds:
            return None
        q = queue.Queue(maxsize=2)
        stop = threading.Event()

        def put(item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def reader():
            try:
                import time
                for i, j in bounds:
                    t0 = time.perf_counter()
                    staged = stage_input_batch(self.args.input_dir, self.graph.network_inputs, shapes, i, j, pinned)
                    self.ingest_s += time.perf_counter() - t0
                    if not put((j - i, staged)):
                        return
            except BaseException as e:  # surfaced in the consumer
                put(e)

        t = threading.Thread(target=reader, daemon=True)
        t.start()
        return q, t, len(bounds), stop

    def close(self):
        """Stops a reader nobody will consume (a run that is dropped before its first sweep)."""
        if self._reader is not None:
            self._reader[3].set()
            self._reader = None

    def _input_batches(self):
        """(b, {input: device tensor}) per batch; the .bin files of the NEXT batches are read into pinned memory by
        a helper thread while the GPU works on the current one.  The first sweep's reader has been running since the run
        was constructed."""
        reader, self._reader = self._reader, None
        if reader is None:
            reader = self._start_reader(self.device.type == "cuda")
            if reader is None:
                return
        q, t, n, stop = reader
        try:
            for _ in range(n):
                item = q.get()
                if isinstance(item, BaseException):
                    raise item
                b, staged = item
                yield b, {name: h.to(self.device, non_blocking=True).reshape(full) for name, (h, full) in staged.items()}
            t.join()
        finally:
            stop.set()

    def forward(self, keep=False):
        """Yields (b, tensors) per batch.  With keep=True the tensor sets stay resident in HBM (up to
        args.resident_gb) so a second pass re-reads them instead of re-running the network."""
        for b, inputs in self._input_batches():
            with self.timed("forward"):
                tensors = self.session.run(inputs)
            mark("first_forward:issued")
            if keep and self._resident_ok:
                nbytes = sum(t.numel() * 4 for t in tensors)
                if self._resident_bytes + nbytes <= self._budget:
                    tensors = self.plan(b).bind(tensors)      # (validated once; pass 2 launches over the held set)
                    self._resident.append((b, tensors))
                    self._resident_bytes += nbytes
                else:
                    self._resident_ok = False
                    self._resident = []
                    self._resident_bytes = 0
            yield b, tensors

    def second_pass(self):
        if self._resident_ok and self._resident:
            yield from self._resident
        else:
            yield from self.forward()

    def release(self):
        self._resident = []
        self._resident_bytes = 0


def _np32(t):
    return t.detach().cpu().numpy().astype(np.float32, copy=False)


def _run_of(onnx_graph, args, run):
    return run if run is not None else CalibrationRun(onnx_graph, args)


def forward_get_minmax(onnx_graph, args, per_image=False, run=None, keep_resident=False):
    """forward_net.py:192-237 — {name: {'max': [...], 'min': [...]}} over this rank's shard."""
    run = _run_of(onnx_graph, args, run)
    if per_image:
        rows = []
        for b, tensors in run.forward(keep=keep_resident):
            plan = run.plan(b)
            acc = ops.CalibAccumulators(plan.n_pairs, run.device)
            acc.minmax_accumulate(plan, tensors, per_image=True)
            lo, hi = acc.finalize_minmax()
            rows.append(torch.stack([lo.reshape(b, run.T), hi.reshape(b, run.T)], -1).clone())
        allr = _np32(torch.cat(rows)) if rows else np.zeros((0, run.T, 2), np.float32)
        return {n: {"max": list(allr[:, t, 1]), "min": list(allr[:, t, 0])} for t, n in enumerate(run.names)}
    acc = ops.CalibAccumulators(run.T, run.device, int(getattr(args, "bins", 2048)))
    # the range kernel of batch i runs on a side stream beside the network forward of batch i + 1 (one is HBM-bound, the
    # other mostly compute-bound library work); a batch's tensors stay referenced until its kernel has run
    main = torch.cuda.current_stream(run.device)
    side = ops._separate_stream(run.device, [main])      # (on a hardware queue of its own: one that shares the forward's runs behind it)
    side.wait_stream(main)                      # the accumulators' initialisation
    in_flight = []
    t_loop = __import__("time").perf_counter()
    for b, tensors in run.forward(keep=keep_resident):
        produced = torch.cuda.Event()
        produced.record(main)
        side.wait_event(produced)
        with torch.cuda.stream(side):
            with run.timed("statistics"):
                acc.minmax_accumulate(run.plan(b), tensors)
            done = torch.cuda.Event()
            done.record(side)
        in_flight.append((done, tensors))
        while len(in_flight) > 2:               # at most two batches ahead of the statistics
            in_flight.pop(0)[0].synchronize()
    main.wait_stream(side)
    in_flight.clear()
    WALL["pass1_loop_s"] = WALL.get("pass1_loop_s", 0.0) + __import__("time").perf_counter() - t_loop
    gmin, gmax = acc.finalize_minmax()
    run.acc = acc
    with wall("results_to_host_s"):
        lo, hi = _np32(gmin), _np32(gmax)
    return {n: {"max": [hi[t]], "min": [lo[t]]} for t, n in enumerate(run.names)}


def _ranges_from_stats(stats_min_max, names, device):
    gmin = np.array([np.min(stats_min_max[n]["min"]) for n in names], np.float32)
    gmax = np.array([np.max(stats_min_max[n]["max"]) for n in names], np.float32)
    return torch.from_numpy(gmin).to(device), torch.from_numpy(gmax).to(device)


def forward_get_hist(onnx_graph, stats_min_max, args, run=None):
    """forward_net.py:240-281 — {name: [int64[bins]]}: the |x| histogram over (0, max(max, -min)) of the
    shard, already summed over images (the reference returns one per image and sums them later,
    basic_algorithm.py:37-38)."""
    run = _run_of(onnx_graph, args, run)
    gmin, gmax = _ranges_from_stats(stats_min_max, run.names, run.device)
    acc = hist_pass(run, gmin, gmax, int(args.bins))
    h = acc.hist.cpu().numpy()
    return {n: [h[t]] for t, n in enumerate(run.names)}


def hist_pass(run, gmin, gmax, bins):
    """Device side of forward_get_hist: install the ranges, sweep the shard, leave the uint64 histograms
    in run.acc.hist.  Raises like np.histogram for a non-finite or degenerate range."""
    acc = getattr(run, "acc", None)
    if acc is None or acc.bins != bins:
        acc = run.acc = ops.CalibAccumulators(run.T, run.device, bins)
    acc.set_minmax(gmin, gmax)
    acc.hist_prepare()
    with wall("pass2_loop_s"):
        for b, tensors in run.second_pass():
            with run.timed("statistics"):
                acc.abs_hist_accumulate(run.plan(b), tensors)
    with wall("release_resident_s"):
        run.release()
    with wall("results_to_host_s"):
        status = acc.range_status()["status"]
    for t, n in enumerate(run.names):
        if status[t] == 1:
            raise ValueError(f"supplied range of [0, {gmax[t].item()}] is not finite (tensor {n})")
        if status[t] == 2:
            raise ValueError(f"Too many bins for data range. Cannot create {bins} finite-sized bins. (tensor {n})")
    return acc


def forward_net_octav(onnx_graph, args, run=None, as_dict=True):
    """forward_net.py:284-342 — {name: {'optimal_s': [...], 'min': [...], 'max': [...]}}, one entry per
    image of the shard.  as_dict=False: the rows stay on the device (run.octav_rows, [n, T, 3]) and nothing is returned — the
    reference's dictionary is 3 T n Python floats, 10 ms of host work per thousand ResNet-50 images that find_clip_val_octav, which
    reads the rows, has no use for."""
    run = _run_of(onnx_graph, args, run)
    dynamic_sym = "dynamic_sym" in platform_setting_table[args.deploy]["qi_params"]
    rows = []
    # (the rescue of batch i runs beside the forward of batch i + 1; the streaming kernel itself stays on this stream, between two
    # forwards: lanes = 1 — overlapped with the next forward's convolutions it slows them by more than it takes, measured)
    pipe = ops.OctavPipeline(dynamic_sym, run.device, lanes=1)
    # (timed: the streaming kernel of every batch — it also walks the pairs — by events on the pipeline's own stream that carries
    # it, beside the next batch's forward; the rescue of the few pairs it could not finish runs on the pipeline's side stream)
    pipe.record_events = bool(getattr(args, "timing_json", None))
    t_loop = __import__("time").perf_counter()
    for b, tensors in run.forward():
        rows.append(pipe.submit(run.plan(b), tensors))
    pipe.sync()
    run._events["statistics"] += pipe.events
    WALL["pass1_loop_s"] = WALL.get("pass1_loop_s", 0.0) + __import__("time").perf_counter() - t_loop
    run.octav_rows = torch.cat(rows) if rows else torch.zeros(0, run.T, 3, device=run.device)
    if not as_dict:
        return None
    with wall("results_to_host_s"):
        r = _np32(run.octav_rows)
    return {n: {"optimal_s": list(r[:, t, 0]), "min": list(r[:, t, 1]), "max": list(r[:, t, 2])}
            for t, n in enumerate(run.names)}


# forward_net.py:345-456 — the reference's "*_transformer" variants compute the same statistics, only walking
# the graph node by node through its host-side ActivationCache to bound host memory.  Here every schedule is
# the batched, HBM-resident one, so they are the same functions.
forward_get_minmax_transformer = forward_get_minmax
forward_get_hist_transformer = forward_get_hist
forward_net_octav_transformer = forward_net_octav


class ActivationCache:
    """Counterpart of forward_net.py:23-190 for callers that want activations by tensor name.

    The reference splits the network into single-node ONNX models, runs one ORT session per node and keeps
    every image's activation of every live tensor on the HOST, evicting by reference count.  Here one batched
    forward of the shard keeps all calibration tensors resident in HBM (288 GB: 1024 ResNet-50 images are
    109 GB).  `cache[name]` returns the list of per-image device tensors (views, no copies), `cache.chunks(name)`
    the per-batch tensors [b, ...] behind them, `cache[initializer]` the initializer array.  `reset()` drops the
    cached activations."""

    def __init__(self, graph, args, st=None, ed=None):
        self.graph, self.args = graph, args
        self.st = 0 if st is None else st
        self.ed = args.data_num if ed is None else ed
        self.activation_cache = {}     # name -> [per-batch device tensors]
        self._filled = False
        self._selective = False        # over the HBM budget: keep only the tensors asked for (one forward each)
        self._sess = None

    def reset(self):
        self.activation_cache.clear()
        self._filled = False

    def _session(self):
        if self._sess is None:
            self._sess = self.graph.make_session(self.args)
        return self._sess

    def _sweep(self, keep):
        """One batched forward of the shard; returns {name: [per-batch tensors]} for the names in `keep`."""
        sess = self._session()
        dev = sess.device if hasattr(sess, "device") else torch.device("cuda", torch.cuda.current_device())
        shapes = {n: self.graph.get_tensor_shape(n) for n in self.graph.network_inputs}
        batch = int(getattr(self.args, "calib_batch", None) or auto_batch(sess.elems_per_image))
        per_name = {n: [] for n in sess.tensor_names if n in keep}
        for i in range(self.st, self.ed, batch):
            j = min(i + batch, self.ed)
            inputs = load_input_batch(self.args.input_dir, self.graph.network_inputs, shapes, i, j, dev)
            for n, t in zip(sess.tensor_names, sess.run(inputs)):
                if n in per_name:
                    per_name[n].append(t)
        return per_name

    def _fill(self):
        """Everything resident if it fits the budget (args.resident_gb, default 160 GB of the 288) — else SELECTIVE mode:
        a tensor's activations are produced when first asked for by one more forward of the shard that keeps only that
        tensor (slower, bounded memory, same values)."""
        sess = self._session()
        need = 4.0 * sum(sess.elems_per_image) * max(0, self.ed - self.st)
        budget = float(getattr(self.args, "resident_gb", 160.0) or 160.0) * 1e9
        self._selective = need > budget
        if self._selective:
            logger.warning("ActivationCache: %.1f GB of activations exceed the %.0f GB budget: keeping tensors on demand",
                           need / 1e9, budget / 1e9)
            self.activation_cache = {}
        else:
            self.activation_cache = self._sweep(set(sess.tensor_names))
        self._filled = True

    def chunks(self, tensor_name):
        if not self._filled:
            self._fill()
        if self._selective and tensor_name not in self.activation_cache:
            one = 4.0 * self._session().elems_per_image[self._session().tensor_names.index(tensor_name)] * (self.ed - self.st)
            held = sum(4.0 * t.numel() for v in self.activation_cache.values() for t in v)
            budget = float(getattr(self.args, "resident_gb", 160.0) or 160.0) * 1e9
            if held + one > budget:
                self.activation_cache.clear()   # make room: the evicted tensors are re-produced if asked for again
            self.activation_cache.update(self._sweep({tensor_name}))
        return self.activation_cache[tensor_name]

    def __getitem__(self, tensor_name):
        if tensor_name in getattr(self.graph, "initializer", {}):
            return self.graph.get_initializer(tensor_name)
        return [t[k] for t in self.chunks(tensor_name) for k in range(t.shape[0])]


def forward_get_tensor(graph, net, index, args):
    """forward_net.py:467-485 — every tensor of calibration image `index`, by name, as device tensors [1, ...]
    (`net`, the reference's ModelProto argument, is not needed: the graph makes its own session)."""
    from collections import OrderedDict
    sess = graph.make_session(args)
    dev = sess.device if hasattr(sess, "device") else torch.device("cuda", torch.cuda.current_device())
    shapes = {n: graph.get_tensor_shape(n) for n in graph.network_inputs}
    inputs = load_input_batch(args.input_dir, graph.network_inputs, shapes, index, index + 1, dev)
    return OrderedDict(zip(sess.tensor_names, sess.run(inputs)))


def log_forward_time(seconds):
    logger.info("Forward time: {:.2f} seconds".format(seconds))


__all__ = ["ActivationSession", "ActivationCache", "CalibrationRun", "input_data_generator", "load_input_batch", "forward_get_minmax",
           "forward_get_hist", "forward_net_octav", "forward_get_tensor", "forward_get_minmax_transformer", "forward_get_hist_transformer",
           "forward_net_octav_transformer", "hist_pass", "DEFAULT_BATCH"]
