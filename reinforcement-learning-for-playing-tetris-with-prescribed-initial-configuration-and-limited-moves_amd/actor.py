"""Actor loop for BASELINE config 5: obs -> policy -> action -> step, entirely on the device.

The policy is this build's own MLP with the reference's sizes -- Linear 217-128-128-128-128-14 with ReLU
(model/model.py:9-20 of the upstream repo, whose constructor does not run as written: SURVEY section 0).  The
dense layers go through torch (hipBLASLt); the environment side (observation expand, action decode, step) is
the HIP library.  One iteration is a fixed sequence of kernel launches on one stream, so it is captured once
into a HIP graph and replayed.
"""
from __future__ import annotations

import torch
from torch import nn

from .env import OBS_DIM, BatchedTetris


class PolicyMLP(nn.Module):
    """Model(217, 14): five Linear layers, ReLU between them (model/model.py:9-20)."""

    def __init__(self, state_space_size: int = OBS_DIM, action_space_size: int = 14, hidden: int = 128):
        super().__init__()
        self.layer1 = nn.Linear(state_space_size, hidden)
        self.layer2 = nn.Linear(hidden, hidden)
        self.layer3 = nn.Linear(hidden, hidden)
        self.layer4 = nn.Linear(hidden, hidden)
        self.layer5 = nn.Linear(hidden, action_space_size)

    def forward(self, x):
        x = torch.relu(self.layer1(x))
        x = torch.relu(self.layer2(x))
        x = torch.relu(self.layer3(x))
        x = torch.relu(self.layer4(x))
        return self.layer5(x)


def policy_image(model: nn.Module, device, f32=False) -> torch.Tensor:
    """The packed image of a PolicyMLP for the fused kernels, on `device`: weights rounded to bf16, (f32=True)
    float32 as they are -- the reference's arithmetic width (model/model.py:9-20) -- or (f32="split") every weight as
    three bf16 pieces: float32 accuracy on the bf16 matrix pipe."""
    from ._lib import pack_policy
    layers = [model.layer1, model.layer2, model.layer3, model.layer4, model.layer5]
    params = [(l.weight.detach().float().cpu().numpy(), l.bias.detach().float().cpu().numpy()) for l in layers]
    return torch.from_numpy(pack_policy(params, f32=f32)).to(device)


class Actor:
    """Greedy actor over a BatchedTetris: buffers are allocated once, the iteration is graph-captured.

    fused=False: observation kernel -> torch Linear layers -> decode kernel -> step kernel.
    fused=True:  one MFMA kernel from the 32-B board state to the action -> step kernel; with dtype=torch.float32 the
                 float32 kernel (csrc/policy_f32.hip) or, split=True, the kernel that reaches float32 accuracy with three
                 bf16 pieces per number (csrc/policy_split.hip, twice as fast); otherwise the bf16 one (csrc/policy_mlp.hip)."""

    def __init__(self, env: BatchedTetris, model: nn.Module, dtype=torch.bfloat16, use_graph: bool = True,
                 fused: bool = False, split: bool = False):
        self.env, self.dtype, self.fused = env, dtype, fused
        if split and not (fused and dtype is torch.float32):
            raise ValueError("split=True goes with fused=True and dtype=torch.float32")
        self.image = policy_image(model, env.device, f32=("split" if split else dtype is torch.float32)) if fused else None
        self.model = model.to(device=env.device, dtype=dtype).eval()
        n, d = env.num_envs, env.device
        self.obs = torch.empty((n, OBS_DIM), dtype=dtype, device=d)
        self.action = torch.empty(n, dtype=torch.uint8, device=d)
        self.reward = torch.empty(n, dtype=torch.float32, device=d)
        self.done = torch.empty(n, dtype=torch.uint8, device=d)
        self._graph = None
        self._use_graph = use_graph

    @torch.no_grad()
    def _iteration(self):
        if self.fused:
            self.env.policy_act(self.image, out=self.action)
            self.env.step_into(self.action, self.reward, self.done)
            return
        self.env.observe(out=self.obs)
        logits = self.model(self.obs)
        self.env.decode_actions(logits.contiguous(), out=self.action)
        self.env.step_into(self.action, self.reward, self.done)

    def _capture(self):
        stream = torch.cuda.Stream(self.env.device)
        stream.wait_stream(torch.cuda.current_stream(self.env.device))
        with torch.cuda.stream(stream):
            for _ in range(2):                       # warm up allocator / hipBLASLt heuristics off-graph
                saved = self.env.snapshot()
                self._iteration()
                self.env.restore(saved)
        torch.cuda.current_stream(self.env.device).wait_stream(stream)
        torch.cuda.synchronize(self.env.device)
        g = torch.cuda.CUDAGraph()
        saved = self.env.snapshot()
        with torch.cuda.graph(g):
            self._iteration()
        self.env.restore(saved)                      # capture does not execute, but keep the state untouched regardless
        self._graph = g
        self._graph_pool = self.env.pool_generation  # the captured launches carry the pool pointers of this moment

    def step(self):
        """One obs -> action -> step iteration; results in self.action / self.reward / self.done."""
        if self._use_graph:
            if self._graph is None or self._graph_pool != self.env.pool_generation:
                self._capture()                      # first use, or load_configs() has replaced the pool since
            self._graph.replay()
            # the library's pool-swap guard counts steps as they pass through its API; a replay does not: tell it
            self.env.note_steps(1)
        else:
            self._iteration()

    def run(self, steps: int):
        for _ in range(steps):
            self.step()
