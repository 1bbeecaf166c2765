"""hipGraph capture of the model forward (torch.cuda.CUDAGraph is a hipGraph on ROCm).

One UNet forward is ~1900 kernel launches; measured on the MI355X box the Python/eager launch path costs ~85 ms of
host time per forward, which made the loop HOST-bound (profiles/r1_bench_trace_summary.txt: the GPU idled ~20 ms per
timestep on 1 GPU and would idle far more once rows are sharded over ranks).  Capturing the forward per batch shape
and replaying it makes the host cost of a phase a handful of launches, so the host RNG work runs far ahead of the GPU.

Static buffers: a hipGraph replays fixed addresses, so each captured shape owns its input tensors (model rows, 0-d
timestep, text / pooled / condition rows) and its output tensor.  The assemble kernels write the model rows straight
into the static input (``input_rows``), so no extra copy is added to the hot loop; side inputs are copied once per image.
"""
import gc
import warnings
import weakref

import torch


class GraphedForward:
    def __init__(self, fwd, enabled=True, warmup_iters=2, prepare=None):
        """``prepare(text, into=None) -> extra`` (optional): values that depend on the side inputs only (the cross-attention
        k / v of the text rows).  It runs eagerly, OUTSIDE the graph, whenever the side inputs are (re)copied -- once per image
        with one image in flight -- and the forward receives its result as a last argument; ``into`` = the result of the
        previous call for this batch shape, to be overwritten in place (the graph reads those tensors at fixed addresses)."""
        self._prepare_ref = None if prepare is None else (weakref.WeakMethod(prepare) if hasattr(prepare, "__self__") else (lambda: prepare))
        # a bound method would make pipeline <-> runner a reference cycle that only the cyclic GC can free -- and a
        # hipGraph being destroyed by a GC pass that happens to run *during another capture* aborts the process
        self._fwd_ref = weakref.WeakMethod(fwd) if hasattr(fwd, "__self__") else (lambda: fwd)
        self.enabled = enabled and torch.cuda.is_available()
        self.warmup_iters = warmup_iters
        self.entries = {}
        self.epoch = 0
        self.replays = 0

    def fwd(self, *a):
        f = self._fwd_ref()
        if f is None:
            raise RuntimeError("the pipeline that owns this GraphedForward is gone")
        return f(*a)

    def prepare(self, text, into=None):
        f = None if self._prepare_ref is None else self._prepare_ref()
        return None if (f is None or text is None) else f(text, into)

    def stats(self):
        """{"captured": shapes replayed as hipGraphs, "eager": shapes whose capture failed (run eagerly), "replays": n}
        -- reported in bench.py's JSON so a silent fallback to eager launches cannot hide a regression."""
        ents = list(self.entries.values())
        return {"enabled": bool(self.enabled), "captured": sum(e["graph"] is not None for e in ents),
                "eager": sum(bool(e["eager"]) for e in ents), "replays": self.replays}

    def new_image(self):
        """Side inputs (text / pooled / condition rows) may have changed: re-copy them on next use."""
        self.epoch += 1

    @staticmethod
    def _key(shape, dtype, cond, t_shape=(), fresh_side=False):
        """One graph per (rows shape, dtype, condition rows, timestep shape, fresh side inputs): the timestep is a 0-d tensor
        when all rows share it and a per-row vector when the rows of several images in flight were fused into one batch."""
        return (tuple(shape), dtype, None if cond is None else (tuple(cond.shape), cond.dtype), tuple(t_shape), bool(fresh_side))

    def input_rows(self, shape, dtype, device, cond=None):
        """The static model-input tensor for this batch shape (allocated on first request; 0-d timestep)."""
        if not self.enabled:
            return torch.empty(shape, dtype=dtype, device=device)
        key = self._key(shape, dtype, cond)
        ent = self.entries.get(key)
        if ent is None:
            ent = {"x": torch.empty(shape, dtype=dtype, device=device), "graph": None, "epoch": -1, "eager": False}
            self.entries[key] = ent
        return ent["x"]

    def _capture(self, ent, t, text, pooled, cond, fresh_side=False):
        x = ent["x"]
        ent["t"] = t.clone()
        ent["text"] = None if text is None else text.clone()
        ent["pooled"] = None if pooled is None else pooled.clone()
        ent["cond"] = None if cond is None else cond.clone()
        # side inputs that change on every call (fused batches of several images in flight) leave nothing to hoist: their
        # k / v projections stay INSIDE the graph (extra = None) instead of ~70 eager launches in front of every replay
        ent["extra"] = None if fresh_side else self.prepare(ent["text"])
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):  # warm-up off the capture: MIOpen find / allocator growth happen here
            for _ in range(self.warmup_iters):
                self.fwd(x, ent["t"], ent["text"], ent["pooled"], ent["cond"], ent["extra"])
        cur.wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        # No Python GC pass may run while the stream is capturing: collecting an old pipeline's hipGraph / memory pool
        # there calls HIP APIs that are illegal during capture and aborts the process (seen on the MI355X box).
        gc.collect()
        gc_was_enabled = gc.isenabled()
        gc.disable()
        try:
            # thread_local: the RCCL watchdog thread polls events while we capture; in "global" mode that would
            # invalidate the capture on multi-GPU runs
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                ent["out"] = self.fwd(x, ent["t"], ent["text"], ent["pooled"], ent["cond"], ent["extra"])
        finally:
            if gc_was_enabled:
                gc.enable()
        ent["graph"] = graph

    def __call__(self, x, t, text=None, pooled=None, cond=None, fresh_side=False):
        """``fresh_side``: the text / pooled / condition rows differ from call to call (fused batches of several images
        in flight), so they are copied into the graph's static buffers on every replay, not once per image."""
        if not self.enabled:
            return self.fwd(x, t, text, pooled, cond, None)
        key = self._key(x.shape, x.dtype, cond, t.shape, fresh_side)
        ent = self.entries.get(key)
        if ent is None:
            ent = {"x": torch.empty_like(x), "graph": None, "epoch": -1, "eager": False}
            self.entries[key] = ent
        if ent["eager"]:
            return self.fwd(x, t, text, pooled, cond, None)
        if x.data_ptr() != ent["x"].data_ptr():
            ent["x"].copy_(x)
        if ent["graph"] is None:
            try:
                self._capture(ent, t, text, pooled, cond, fresh_side)
            except Exception as e:  # noqa: BLE001 -- a library that cannot be captured must not take the path down
                warnings.warn(f"hipGraph capture failed for rows {tuple(x.shape)} ({type(e).__name__}: {e}); "
                              "running this shape eagerly")
                ent["eager"] = True
                torch.cuda.synchronize()
                return self.fwd(x, t, text, pooled, cond, None)
            ent["epoch"] = self.epoch
        ent["t"].copy_(t)
        if fresh_side or ent["epoch"] != self.epoch:
            for name, src in (("text", text), ("pooled", pooled), ("cond", cond)):
                if src is not None:
                    ent[name].copy_(src)
            if ent.get("extra") is not None:
                self.prepare(ent["text"], ent["extra"])   # same tensors, new contents
            ent["epoch"] = self.epoch
        ent["graph"].replay()
        self.replays += 1
        return ent["out"]
