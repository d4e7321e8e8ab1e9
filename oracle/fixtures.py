"""Helpers shared by the fixture generator (oracle/make_golden.py, which imports the reference and runs only in the build
container) and by the tests that replay its fixtures (which must not import that module on the GPU box): nothing here
touches /root/reference.  Test infrastructure only."""
import numpy as np

# keyword arguments of the feature-inversion generator, inversion.py:21-25 (positional: input depth, 3 output channels)
INVERSION_NET = dict(num_channels_down=[16, 32, 64, 128, 128, 128], num_channels_up=[16, 32, 64, 128, 128, 128],
                     num_channels_skip=[4, 4, 4, 4, 4, 4], filter_size_down=[7, 7, 5, 5, 3, 3], filter_size_up=[7, 7, 5, 5, 3, 3],
                     downsample_mode='stride', pad='reflection')


def sample(t, n=257):
    """Deterministic strided subsample of a tensor (keeps fixtures small)."""
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy().copy()


def stats(t):
    """(sum, sum |.|, sum of squares) in fp64."""
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()], np.float64)
