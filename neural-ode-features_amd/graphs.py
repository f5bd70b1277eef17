"""hipGraph capture of the parts of a step whose launch sequence is fixed: the stem and the classifier head.  The ODE solves between
them stay stream-ordered launches (their step count is data dependent; a resident bs = 1 solve carries a nonce a replay would repeat).

* `capture_inference(model, sample_x)` (round 6): the bs = 1 census of the reference (`evaluate.py:97-142`, every test image solved
  on its own) spends as long in the stem's eleven launches, the head's four and the Python between them as in the ODE block's one
  resident launch.  Under `model.eval()` + `torch.no_grad()` and for inputs of the captured shape, `model(x)` becomes: one copy into
  the stem's static input, ONE graph replay (stem), the block's solve, one copy, ONE graph replay (head + Linear).  Numbers:
  `profiles/r06_census_bs1.txt` (`tools/census_bs1.py --graphs`).
* `capture_static_parts(model, sample_x)`: forward AND backward of stem / head through `torch.cuda.make_graphed_callables` for
  training.  `bench.py --graphs` captures the head only: the stem's backward graph replays on the capture's side stream and the eager
  optimizer behind it pays a cross-stream synchronisation per parameter (measured slower; history r1-r3).  Off by default.

Capture BEFORE `torch.distributed.init_process_group` (RCCL's watchdog polls events, which a capturing stream forbids).  Parameters
keep their storages (in-place updates, `load_state_dict` and `dp.broadcast_parameters` are seen by the graphs); shapes are frozen.
"""
from __future__ import annotations

import torch
from torch import nn


def capture_static_parts(model: nn.Module, sample_x: torch.Tensor, stem: bool = True, head: bool = True) -> nn.Module:
    """Replace `model.downsample` / `model.classifier` by graphed callables (training mode; `model.eval()` falls
    back to the eager modules).  `sample_x`: a batch of the shape the model will be trained on."""
    if not sample_x.is_cuda:
        raise RuntimeError('graph capture needs a HIP device')
    was_training = model.training
    model.train()
    with torch.no_grad():
        h = model.downsample(sample_x)
    if isinstance(h, (tuple, list)):
        raise NotImplementedError('ODE stems return trajectories in feature-extraction mode; capture the plain stems only')
    if stem and not any(isinstance(m, _ode_block_type()) for m in model.downsample.modules()):
        x = sample_x.detach().clone()          # images need no gradient: the backward graph stops at the first conv's weights
        model.downsample = torch.cuda.make_graphed_callables(model.downsample, (x,))
    if head:
        hs = torch.randn_like(h).requires_grad_(True)
        model.classifier = torch.cuda.make_graphed_callables(model.classifier, (hs,))
    model.train(was_training)
    return model


class _InferenceGraphs:
    """Static buffers + two captured graphs (stem; head) of one input shape, eval / no_grad only."""

    def __init__(self, model: nn.Module, sample_x: torch.Tensor):
        self.shape, self.dtype, self.device = tuple(sample_x.shape), sample_x.dtype, sample_x.device
        self.x = sample_x.detach().clone()
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(3):                                   # warm-up outside the capture: workspaces, lazy library state
                h = model.downsample(self.x)
                p = model.classifier(h)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        self.g_stem, self.g_head = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.no_grad():
            with torch.cuda.graph(self.g_stem):
                self.h = model.downsample(self.x)
            self.hin = torch.empty_like(self.h)
            with torch.cuda.graph(self.g_head, pool=self.g_stem.pool()):
                self.p = model.classifier(self.hin)

    def matches(self, x: torch.Tensor) -> bool:
        return tuple(x.shape) == self.shape and x.dtype == self.dtype and x.device == self.device


def capture_inference(model: nn.Module, sample_x: torch.Tensor) -> nn.Module:
    """Capture `model.downsample` and `model.classifier` as inference graphs for inputs shaped like `sample_x` (plain stems only).
    Afterwards `model(x)` under `model.eval()` + `torch.no_grad()` replays them around the ODE block's own launches; any other call
    (training, another shape, gradients wanted) takes the ordinary path.  Returns the model (its class is swapped for a subclass)."""
    if not sample_x.is_cuda:
        raise RuntimeError('graph capture needs a HIP device')
    if any(isinstance(m, _ode_block_type()) for m in model.downsample.modules()):
        raise NotImplementedError('ODE stems hold a solve: capture the plain stems only')
    was_training = model.training
    model.eval()
    graphs = _InferenceGraphs(model, sample_x)
    model.train(was_training)
    base = type(model)

    class _Graphed(base):
        def forward(self, x):
            g = self.__dict__.get('_inference_graphs')
            if g is None or self.training or torch.is_grad_enabled() or not g.matches(x):
                return base.forward(self, x)
            g.x.copy_(x)
            g.g_stem.replay()
            y = self.odeblock(g.h)
            if y.dim() != 4:
                return torch.stack([self.classifier(yi) for yi in y])
            g.hin.copy_(y)
            g.g_head.replay()
            return g.p          # (a static buffer: overwritten by the next call -- read it, or clone it, before that)

    _Graphed.__name__ = base.__name__
    model.__class__ = _Graphed
    model.__dict__['_inference_graphs'] = graphs
    return model


def _ode_block_type():
    from .modules import ODEBlock
    return ODEBlock
