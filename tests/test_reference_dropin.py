"""Drop-in check of the boundary (SURVEY.md §8-b "who calls it"): the reference's OWN consumers of the model API —
Util/network_util.py (Get_Network_Shape, Build_Generator_From_Dict, Get_Generator_Styles), Util/mask_util.py
(Mask_the_Generator) and Util/pruning_util.py (Get_Uniform_RmveList, Generate_Prune_Mask_List) — executed unmodified
from the read-only reference checkout with `model` / `op` resolving to THIS package.  Runs only where the reference is
mounted (the build container); nothing of the reference is copied or travels."""
import os
import sys
import types

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "Util")), reason="reference checkout not mounted")


@pytest.fixture(scope="module")
def ref_utils():
    import model as product_model   # tests/conftest.py puts content-aware-gan-compression_amd first on sys.path
    assert "content-aware-gan-compression_amd" in product_model.__file__
    saved = {k: sys.modules.get(k) for k in ("torchvision", "torchvision.utils", "PIL", "PIL.Image", "Util")}
    for name in ("torchvision", "PIL"):     # not installed here and not used by the functions under test
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules.setdefault("torchvision.utils", types.ModuleType("torchvision.utils"))
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    sys.modules.setdefault("PIL.Image", types.ModuleType("PIL.Image"))
    sys.modules["PIL"].Image = sys.modules["PIL.Image"]
    sys.path.append(REF)                     # AFTER the product: only `Util` is found there
    old_bytecode = sys.dont_write_bytecode
    sys.dont_write_bytecode = True           # nothing is written into the read-only checkout (no __pycache__)
    try:
        from Util import mask_util, network_util, pruning_util
        assert network_util.Generator is product_model.Generator   # the reference code is driving the product class
        yield network_util, mask_util, pruning_util
    finally:
        sys.dont_write_bytecode = old_bytecode
        sys.path.remove(REF)
        for k in [m for m in sys.modules if m == "Util" or m.startswith("Util.")]:
            del sys.modules[k]
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)


def test_reference_consumers_run_unchanged_on_the_product_generator(ref_utils):
    network_util, mask_util, pruning_util = ref_utils
    import model as M
    from cagc import prune
    torch.manual_seed(0)
    full = M.Generator(64, 32, 2)
    sd = full.state_dict()
    shape = network_util.Get_Network_Shape(sd)                      # reference parser on the product's key names/order
    assert shape == prune.network_shape(sd) == [512] * 10
    rmve = pruning_util.Get_Uniform_RmveList(shape, 0.7)
    assert [int(v) for v in rmve] == [int(v) for v in prune.uniform_remove_list(shape, 0.7)]
    rng = np.random.RandomState(1)
    scores = [rng.rand(c) for c in shape]
    masks_ref = pruning_util.Generate_Prune_Mask_List([torch.tensor(s) for s in scores], shape, rmve, info_print=False)
    masks = prune.masks_from_scores(scores, shape, rmve)
    assert all(np.array_equal(np.asarray(a, dtype=bool), np.asarray(b, dtype=bool)) for a, b in zip(masks_ref, masks))
    pruned_ref = mask_util.Mask_the_Generator(sd, masks_ref)        # reference surgery on the product's state dict
    pruned = prune.mask_generator_state_dict(sd, masks)
    assert list(pruned_ref) == list(pruned)
    assert all(torch.equal(pruned_ref[k], pruned[k]) for k in pruned)
    student = network_util.Build_Generator_From_Dict(pruned_ref, size=64, latent=32, n_mlp=2)   # builds the PRODUCT class
    assert isinstance(student, M.Generator) and network_util.Get_Network_Shape(student.state_dict()) == [154] * 10
    z = torch.randn(2, 32)
    with torch.no_grad():
        img = student([z], randomize_noise=False)
        styles = network_util.Get_Generator_Styles(student, z)      # reads .style / .conv1.conv.modulation / .convs / .to_rgbs
    assert tuple(img.shape) == (2, 3, 64, 64) and torch.isfinite(img).all()
    assert len(styles) == 10 and all(np.isfinite(s).all() for s in styles) and styles[0].shape == (2, 154)
