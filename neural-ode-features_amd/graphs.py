"""hipGraph capture of the parts of the training step whose launch sequence is fixed: the stem and the classifier
head (forward AND backward, through `torch.cuda.make_graphed_callables`).  The ODE solves between them stay
stream-ordered launches (their step count is data dependent; see csrc/node_api.hip `run_steps`).

Why: after each solve the host has just synchronised with the GPU, so everything it dispatches next is exposed --
~30 eager launches for the stem's backward, ~15 for head + loss.  Replaying a captured graph is one launch each
(measured at cfg 2, tools/phase_times.py: head + loss 0.22 -> 0.09 ms, head backward 0.35 -> 0.21 ms per step; whole
step, same-process A/B with tools/ab_step.py: 8.93 -> 8.80 ms with the head captured.  Capturing the stem made the
step SLOWER, 9.16 ms: its backward graph replays on the capture's side stream and the eager optimizer behind it
pays a cross-stream synchronisation per parameter -- so `bench.py` captures the head only).

Capture BEFORE `torch.distributed.init_process_group`: RCCL's watchdog thread polls events, which is not allowed
while a stream of the process is capturing.  Parameters keep their storages (in-place updates, `load_state_dict`
and `dp.broadcast_parameters` are seen by the graphs); shapes are frozen to the sample's.
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


def _ode_block_type():
    from .modules import ODEBlock
    return ODEBlock
