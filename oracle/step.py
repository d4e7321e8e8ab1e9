"""One reference-shaped optimisation step, ``train.py:51-80`` (test infra / CPU baseline).

Runs exactly what the reference executes per step: Model.forward
(``models/model.py:12-25``: G on A_global, on A every 75th step, on B_global), LossG
(6 ViT forwards ordinary step / 10 on every 75th), autograd backward (3 / 5 ViT
backwards), Adam.  ViT parameters are frozen here (the reference leaves
``requires_grad`` on and wastes the weight-grads; results are identical)."""
import torch

from . import generator as gen
from . import losses as L
from .optim import Adam


class SpliceOracle:
    def __init__(self, vit, gen_params, cfg):
        self.vit = vit
        for p in self.vit.parameters():
            p.requires_grad_(False)
        self.cfg = dict(L.DEFAULT_CFG, **cfg)
        self.params = {k: v.clone().requires_grad_(True) for k, v in gen_params.items()}
        self.lambdas = L.initial_lambdas(self.cfg)
        c = self.cfg
        self.opt = Adam(self.params.values(), c["lr"], c["optimizer_beta1"], c["optimizer_beta2"])
        self.step_idx = -1  # data/Dataset.py:57 -- first __getitem__ makes it 0
        self.grad_hook = None   # test infrastructure (oracle/trajectory_ensemble.py): grads -> grads before the optimizer step

    def model_forward(self, inputs):
        c = self.cfg
        out = {}
        if c["lambda_global_cls"] + c["lambda_global_ssim"] > 0:
            out["x_global"] = gen.forward(self.params, inputs["A_global"])
        if c["lambda_entire_ssim"] > 0 and inputs["step"] % c["entire_A_every"] == 0:
            out["x_entire"] = gen.forward(self.params, inputs["A"])
        out["y_global"] = gen.forward(self.params, inputs["B_global"])
        return out

    def step(self, A_global, B_global, A_entire=None):
        """A_global/B_global: ``[n_crops,3,s,s]``; A_entire ``[1,3,H,W]`` (needed when
        step % entire_A_every == 0).  Returns (losses dict of floats, outputs)."""
        self.step_idx += 1
        inputs = {"step": self.step_idx, "A_global": A_global, "B_global": B_global}
        if self.step_idx % self.cfg["entire_A_every"] == 0:
            inputs["A"] = A_entire
        outputs = self.model_forward(inputs)
        losses = L.loss_g(self.vit, self.cfg, self.lambdas, outputs, inputs)
        grads = torch.autograd.grad(losses["loss"], list(self.params.values()), allow_unused=True)
        grads = [torch.zeros_like(p) if g is None else g for g, p in zip(grads, self.params.values())]
        if self.grad_hook is not None:
            grads = self.grad_hook(grads)
        self.opt.step(grads)
        return {k: float(v.detach()) for k, v in losses.items()}, outputs, grads
