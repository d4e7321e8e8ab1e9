"""Drop-in for ``models/model.py``: ``Model(cfg)`` with ``.netG`` and ``forward(input) -> dict``.

The reference's ``Model`` is the object the training loop asks "which generator calls does this step need?"
(``models/model.py:12-25``).  Here that question is answered by a route table, and the same object can hand its
generator over to the fused HIP step (``fused_engine``): the façade path (one autograd op per generator call) and the
fused path (``SpliceEngine``: one host call per optimisation step) then share ONE parameter arena, so a notebook may
mix ``model(inputs)`` calls with fused steps and read ``model.netG.state_dict()`` at any time.
"""
import torch

from . import networks

# (key of the returned dict, key of the input batch it is generated from, predicate(cfg, input) -> wanted this step?)
# in the order the reference fills its dict.
_ROUTES = (
    ("x_global", "A_global", lambda cfg, inp: cfg['lambda_global_cls'] + cfg['lambda_global_ssim'] > 0),
    ("x_entire", "A", lambda cfg, inp: cfg['lambda_entire_ssim'] > 0 and inp['step'] % cfg['entire_A_every'] == 0),
    ("y_global", "B_global", lambda cfg, inp: True),   # generated every step, even when the identity term is off
)


class Model(torch.nn.Module):
    def __init__(self, cfg):
        super().__init__()
        if not torch.cuda.is_available():
            raise RuntimeError("Model: the HIP generator needs an MI355X; there is no CPU fallback")
        self.cfg = cfg
        self.netG = networks.define_G(cfg['init_type'], cfg['init_gain'], device=torch.device('cuda'))

    def forward(self, input):
        return {out_key: self.netG(input[in_key]) for out_key, in_key, wanted in _ROUTES if wanted(self.cfg, input)}

    def fused_engine(self, vit_state, crop_hw, entire_hw=None, vit_engine=None):
        """A ``SpliceEngine`` whose parameter arena IS ``netG``'s arena (no copy): fused steps update the weights the
        façade sees.  ``crop_hw`` / ``entire_hw`` as in ``SpliceEngine``."""
        from .engine import SpliceEngine
        state = {k: v.detach() for k, v in self.netG.state_dict().items() if k in self.netG.engine.table}
        eng = SpliceEngine(self.cfg, vit_state, state, crop_hw, entire_hw, device=self.netG.flat.device, vit_engine=vit_engine)
        eng.params = self.netG.flat            # share, do not copy: nn.Parameters are views of this tensor
        return eng
