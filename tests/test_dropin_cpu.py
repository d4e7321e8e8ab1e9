"""The reference's own import lines resolve to the engine after `import splice_amd.dropin` (SURVEY §8b:
`train.py:4-7`, `Splice.ipynb` cell 8, `inversion.py:1-4`, `keys_self_sim_pca.py:1`).  Runs in a child interpreter so the aliases
(`util`, `data`, `train` ...) never leak into the test session."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code):
    env = dict(os.environ, PYTHONPATH=ROOT)
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], capture_output=True, text=True, env=env, cwd="/tmp", timeout=300)


def test_reference_import_lines_bind_engine_objects():
    r = _run("""
        import splice_amd.dropin
        # Splice.ipynb cell 8
        from train import train_model
        # train.py:4-7
        from data.Dataset import SingleImageDataset
        from models.model import Model
        from util.losses import LossG
        from util.util import get_scheduler, get_optimizer, save_result
        # inversion.py:1,4 / keys_self_sim_pca.py:1
        from models.extractor import VitExtractor, attn_cosine_sim
        from models.unet.skip import skip
        from models.networks import define_G, init_weights, init_net
        from util.util import tensor2im
        from data.transforms import Global_crops, dino_structure_transforms, dino_texture_transforms
        import models.extractor, models.networks, util.losses
        import splice_amd.train, splice_amd.extractor, splice_amd.model, splice_amd.losses, splice_amd.util, splice_amd.networks
        assert train_model is splice_amd.train.train_model
        assert VitExtractor is splice_amd.extractor.VitExtractor and attn_cosine_sim is splice_amd.extractor.attn_cosine_sim
        assert models.extractor.VitExtractor.KEY_LIST == ['block', 'attn', 'patch_imd', 'qkv']
        assert Model is splice_amd.model.Model and LossG is splice_amd.losses.LossG
        assert get_optimizer is splice_amd.util.get_optimizer and save_result is splice_amd.util.save_result
        assert skip is splice_amd.networks.skip and define_G is splice_amd.networks.define_G
        assert models.networks.define_G is define_G and util.losses.LossG is LossG
        import inversion, keys_self_sim_pca
        assert inversion.__name__ == 'splice_amd.inversion' and keys_self_sim_pca.__name__ == 'splice_amd.keys_self_sim_pca'
        # idempotent, and removable
        assert splice_amd.dropin.install() and 'train' in __import__('sys').modules
        splice_amd.dropin.uninstall()
        assert 'train' not in __import__('sys').modules and 'models.extractor' not in __import__('sys').modules
        print('ok')
    """)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_foreign_modules_are_not_silently_replaced(tmp_path):
    (tmp_path / "train.py").write_text("def train_model(*a):\n    return 'foreign'\n")
    code = """
        import os, sys
        sys.path.insert(0, FOREIGN)
        import train                       # somebody else's `train` is already imported
        try:
            import splice_amd.dropin
        except ImportError as e:
            assert 'train' in str(e), e
        else:
            raise SystemExit('expected ImportError')
        assert train.train_model() == 'foreign' and sys.modules['train'] is train
        os.environ['SPLICE_DROPIN_FORCE'] = '1'      # explicit override
        import splice_amd.dropin, splice_amd.train
        from train import train_model
        assert train_model is splice_amd.train.train_model
        print('ok')
    """.replace("FOREIGN", repr(str(tmp_path)))
    r = _run(code)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_single_image_dataset_contract(tmp_path):
    """`dataset[0]` / `get_A()` / `len` as train.py:34,53-55,71 use them (tensors only in the sample; 'A' every 75th step)."""
    import numpy as np
    from PIL import Image
    for side, hw in (("A", (40, 56)), ("B", (48, 36))):
        os.makedirs(tmp_path / side)
        Image.fromarray((np.random.RandomState(1).rand(hw[0], hw[1], 3) * 255).astype(np.uint8)).save(tmp_path / side / "img.png")
    r = _run(f"""
        import torch, splice_amd.dropin
        import splice_amd.train as T
        T.device = torch.device('cpu')          # the feed itself is device-agnostic; the engine is what needs the GPU
        from data.Dataset import SingleImageDataset
        cfg = dict(dataroot={str(tmp_path)!r}, A_resize=0, B_resize=0, direction='AtoB', use_augmentations=False, entire_A_every=75,
                   global_A_crops_n_crops=1, global_B_crops_n_crops=2, global_A_crops_min_cover=0.95, global_B_crops_min_cover=0.95)
        ds = SingleImageDataset(cfg)
        assert len(ds) == 1 and ds.get_A().shape == (1, 3, 40, 56)
        for i in range(77):
            s = ds[0]
            assert all(torch.is_tensor(v) for v in s.values())
            assert float(s['step']) == i and ('A' in s) == (i % 75 == 0)
            assert s['A_global'].shape[:2] == (1, 3) and s['B_global'].shape[:2] == (2, 3)
            assert 38 <= s['A_global'].shape[2] == s['A_global'].shape[3] <= 40 and s['B_global'].shape[2] == 36
        print('ok')
    """)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_install_is_idempotent_and_uninstall_restores_foreign_modules():
    """a second install() must bind the SAME module objects; install(force=True) over a foreign ``train`` module must be undone
    by uninstall().  Child interpreter: the aliases are process-global."""
    code = r"""
import sys, types
import splice_amd.dropin as d
import train as t1
skip = sys.modules['models.unet.skip']
d.install()
assert sys.modules['models.unet.skip'] is skip and sys.modules['train'] is t1 and sys.modules['data.Dataset'] is sys.modules['data'].Dataset
d.uninstall()
assert 'train' not in sys.modules
foreign = types.ModuleType('train')
sys.modules['train'] = foreign
try:
    d.install()
    raise SystemExit('install() over a foreign module must raise')
except ImportError:
    pass
d.install(force=True)
assert sys.modules['train'] is not foreign
d.uninstall()
assert sys.modules['train'] is foreign
print('OK')
"""
    import subprocess, sys as _sys, os as _os
    root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    r = subprocess.run([_sys.executable, "-c", code], capture_output=True, text=True, cwd=root, env=dict(_os.environ, PYTHONPATH=root), timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.stdout, r.stderr)
