"""hipGraph replay of the forward path (MI355X: "HIP graphs instead of a tracing compiler").

RoseTTAFold.forward issues ~4,800 launches through the C ABI on the current stream; none of them depends on a host read
of device data (the kNN edge count and the input validation flags stay on the device), so one stream capture records the
whole forward into ONE hipGraph.  Measured (tools/hipgraph_capture.py, profiles/r03_hipgraph_capture.log): at the
benchmark configuration the GPU is the bottleneck either way (326.4 ms eager, 324.1 ms replay; the host needs 128 ms to
issue the launches), at small shapes the forward is launch-bound and replay is 2.8x faster (config 1: 12.4 -> 4.5 ms).
Replays are bitwise equal to the eager forward.

    g = GraphedForward(model, msa, seq, aa_idx)      # captures (the example inputs fix the shapes)
    logits, xyz, plddt = g(msa2, seq2, aa_idx2)      # validates, copies into the static inputs, replays

The returned tensors are the graph's static outputs: they are overwritten by the next call (clone what must survive).
Validation (token / residue-index range -> IndexError, as the reference's nn.Embedding raises) runs before every replay,
outside the graph: it needs a 12-byte read-back.  The graph's kernels read the prepared 16-bit weight copies by raw
pointer, so every call compares the parameters' fingerprint (model.weights_fingerprint: one pass over the parameter list,
what the public forward does as well) with the one of the capture and records a new graph when load_state_dict /
load_checkpoint / .to() / an in-place parameter edit / invalidate_weight_caches happened in between.
"""
import torch

from . import _lib as L
from . import model as M
from . import structure as S


class GraphedForward:
    def __init__(self, model, msa, seq, aa_idx, warmup=2):
        if not msa.is_cuda:
            raise L.RfmiError("GraphedForward needs device tensors; there is no CPU fallback")
        self.model = model
        self.device = msa.device
        self._in = tuple(t.detach().clone().contiguous() for t in (msa, seq, aa_idx))
        self._warmup = warmup
        self.recapture()

    def _validate(self, msa, seq, aa_idx):
        m = self.model
        return M.check_index_range(msa, seq, aa_idx, m.msa_emb.to_embedding.num_embeddings,
                                   min(m.msa_emb.pos_enc.max_len, m.pair_emb.pos_enc.max_len))

    def recapture(self, monotonic=None):
        """(Re)record the graph: after load_state_dict / set_compute_dtype, or for the other residue-index branch."""
        if self.model.training:
            raise L.RfmiError("GraphedForward records the inference forward: the dropout masks of a training-mode forward are chosen on "
                              "the host per call (model.manual_seed / counters), a replay would repeat one set of masks; call model.eval()")
        with torch.cuda.device(self.device), torch.no_grad():
            self._dtype = M.T()   # the graph holds the kernels of the library active now
            # the recorded kernels read the prepared 16-bit weight copies by raw pointer: bring the copies in line with the live
            # parameters first (what the public model(...) call does), and remember which parameters the graph belongs to
            fp = M.weights_fingerprint(self.model)
            if fp != self.model._rf_fp:
                M.invalidate_weight_caches(self.model)
                object.__setattr__(self.model, "_rf_fp", fp)
            self._fp = fp
            self._epoch = M.RT.cache_epoch
            self._mono = self._validate(*self._in) if monotonic is None else monotonic
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(self._warmup):      # weight copies and workspaces are prepared outside the capture
                    self.model.forward_validated(*self._in, self._mono)
                side.synchronize()
                self.graph = torch.cuda.CUDAGraph()
                S._PENDING_EDGE_COUNTS.clear()
                # thread-local capture mode: HIP calls of OTHER threads (the RCCL watchdog of a torch.distributed process group
                # polling its events) must not invalidate this thread's capture
                with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
                    self._out = self.model.forward_validated(*self._in, self._mono)
                self._edge_counts = list(S._PENDING_EDGE_COUNTS)   # static tensors of the graph: checked after every replay
                S._PENDING_EDGE_COUNTS.clear()
            torch.cuda.current_stream().wait_stream(side)

    def __call__(self, msa, seq, aa_idx):
        for new, static in zip((msa, seq, aa_idx), self._in):
            if new.shape != static.shape or new.dtype != static.dtype or new.device != static.device:
                raise ValueError(f"GraphedForward was captured for {tuple(static.shape)} {static.dtype} on {static.device}; "
                                 f"got {tuple(new.shape)} {new.dtype} on {new.device}")
        if self.model.training:
            raise L.RfmiError("GraphedForward replays the inference forward; the model is in training mode (model.eval() first)")
        if M.T() != self._dtype:
            raise L.RfmiError(f"GraphedForward was captured in {self._dtype} mode; set_compute_dtype changed it to {M.T()}: call recapture()")
        with torch.cuda.device(self.device):
            msa, seq, aa_idx = msa.contiguous(), seq.contiguous(), aa_idx.contiguous()
            mono = self._validate(msa, seq, aa_idx)   # IndexError before anything is overwritten
            # weights changed since the capture (load_state_dict / load_checkpoint / .to() / p.copy_ / invalidate_weight_caches):
            # the graph's kernels point at weight copies that were freed or are stale -> record a new graph
            stale = M.weights_fingerprint(self.model) != self._fp or M.RT.cache_epoch != self._epoch
            if mono != self._mono or stale:                    # the structure track takes another branch for unordered residue indices
                for new, static in zip((msa, seq, aa_idx), self._in):
                    static.copy_(new)
                self.recapture(monotonic=mono)
            else:
                for new, static in zip((msa, seq, aa_idx), self._in):
                    static.copy_(new)
            self.graph.replay()
            S.check_edge_capacity(self._edge_counts)
        return self._out
