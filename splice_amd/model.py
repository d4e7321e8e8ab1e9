"""Drop-in for ``models/model.py``: ``Model(cfg).netG`` and ``forward(input) -> dict``."""
import torch

from . import networks


class Model(torch.nn.Module):
    def __init__(self, cfg):
        super().__init__()
        device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
        self.netG = networks.define_G(cfg['init_type'], cfg['init_gain'], device=device)
        self.cfg = cfg

    def forward(self, input):
        """``models/model.py:12-25`` verbatim in behaviour: x_global, x_entire (every
        ``entire_A_every``-th step), y_global (always)."""
        outputs = {}
        if self.cfg['lambda_global_cls'] + self.cfg['lambda_global_ssim'] > 0:
            outputs['x_global'] = self.netG(input['A_global'])
        if self.cfg['lambda_entire_ssim'] > 0 and input['step'] % self.cfg['entire_A_every'] == 0:
            outputs['x_entire'] = self.netG(input['A'])
        outputs['y_global'] = self.netG(input['B_global'])
        return outputs
