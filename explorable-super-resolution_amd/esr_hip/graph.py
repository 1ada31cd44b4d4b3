"""HIP-graph replay of an inference forward (no autograd).  The RRDB-23 x4 + CEM forward is ~360 kernel launches; GraphedForward
captures them once per input shape with torch.cuda.CUDAGraph (a hipGraph on ROCm) and replays the graph on later calls, which takes
the host out of the loop (useful when the Python thread is busy, e.g. inside the GUI's event loop).

Measured on MI355X (round 1): replay is bit-identical to eager and NOT faster on an idle host — RRDB-23 on one 32x32 crop 8.91 ms
eager vs 8.96 ms replayed, RRDB-3 1.42 vs 1.43 ms: at small sizes each conv launch is bound by its own serial K loop (a handful of
workgroups, ~25 us each), not by launch overhead, and the launches already queue back to back.

    fast = GraphedForward(netG)            # netG: RRDBNet or the CEM-wrapped generator, in eval mode, on the GPU
    y = fast(x)                            # first call per shape: warm-up + capture; later calls: copy-in, one graph launch

The captured graph bakes in the activation buffers and the PACKED weights of the moment of capture; it is dropped and re-captured
when any parameter changes (optimizer step, load_state_dict).  The returned tensor is a buffer owned by the graph: clone it if it
must survive the next call.
"""
import torch

from ._lib import EsrError


class GraphedForward:
    def __init__(self, module, max_graphs=4):
        self.module, self.max_graphs = module, max_graphs
        self._graphs = {}

    def _param_key(self):
        # (storage, version) of every parameter plus the engines' invalidation counters: `.data` edits do not bump a version (see
        # RRDBEngine.invalidate), an explicit invalidate() does bump the counter
        engines = tuple(m.engine.generation for m in self.module.modules() if hasattr(m, 'invalidate_packs'))
        return tuple((p.data_ptr(), p._version) for p in self.module.parameters()) + engines

    def __call__(self, x):
        if not x.is_cuda or (torch.is_grad_enabled() and x.requires_grad):
            raise EsrError('GraphedForward is an inference path: it takes a GPU tensor that does not require grad')
        key = (tuple(x.shape), x.dtype, x.device.index)
        entry = self._graphs.get(key)
        pkey = self._param_key()
        if entry is None or entry['pkey'] != pkey:
            if len(self._graphs) >= self.max_graphs:
                self._graphs.clear()
            static_x = x.detach().clone()
            with torch.no_grad():
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):                       # warm-up off the capture: packs weights, creates the cached buffers,
                    for _ in range(2):                           # sets kernel attributes (none of that may happen inside a capture)
                        self.module(static_x)
                torch.cuda.current_stream().wait_stream(s)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    static_y = self.module(static_x)
            entry = dict(graph=g, x=static_x, y=static_y, pkey=pkey)
            self._graphs[key] = entry
        entry['x'].copy_(x)
        entry['graph'].replay()
        return entry['y']

    def check_range(self):
        """fp16 precisions: raise EsrError if a replayed forward stored saturated activations (RRDBEngine.check_range; a replay cannot post the
        range word to the host by itself, so this reads it from the device — call it where an image is handed on)."""
        for m in self.module.modules():
            if hasattr(m, 'invalidate_packs') and hasattr(m, 'check_range'):
                m.check_range(wait=True)
