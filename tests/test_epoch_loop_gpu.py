"""The product loop as the reference's `_run_training` drives it (model_handler.py:264-285), with REAL torch DataLoaders over a
WSIPatch-shaped dataset (dataset/PatchWSI.py:65-83: `(index [1] int, (feats [N, 1024], ext), label [2])`, batch_size 1, shuffle,
worker processes, no pin_memory): two epochs of `_train_each_epoch` + `test_model` on a validation loader after each. Checks the
ingest end to end -- pageable worker tensors through the copy pool and the staging slab, the device-wide bag cache scoped by the
DATASET object (second epoch: hits only, for the shuffled training loader and the sequential validation loader alike), the slab
pad, the batched evaluation -- and that a second handler over another dataset does not see the first one's bags."""
import pytest
import torch
from torch.utils.data import DataLoader, Dataset

from advmil_amd import synth
from advmil_amd.config import default_cfg
from tests import helpers as H
from tests.test_parity_gpu import DEV, load_synth

pytestmark = pytest.mark.gpu


class Patients(Dataset):
    def __init__(self, first, lens, ratio_mask=None):
        self.first, self.lens, self.ratio_mask = first, lens, ratio_mask

    def __len__(self):
        return len(self.lens)

    def __getitem__(self, index):
        n = self.lens[index]
        feats = H.T(synth.bag(H.DATA_SEED, self.first + index, n))[0]
        label = H.label(self.first + index)[0]
        return torch.Tensor([index]).to(torch.int), (feats, torch.Tensor([0])), label


@pytest.mark.timeout(600)
def test_two_epochs_with_torch_dataloaders():
    from advmil_amd import ingest
    from advmil_amd.model import MyHandler
    train_ds = Patients(700, (1040, 2064, 528, 1536, 2048, 784, 1296, 912, 1808, 640, 1120, 2000))       # 12 bags, 4 per step
    val_ds = Patients(800, (1024, 560, 1904, 1312, 720))
    gen = torch.Generator().manual_seed(3)
    train = DataLoader(train_ds, batch_size=1, shuffle=True, num_workers=2, generator=gen)
    val = DataLoader(val_ds, batch_size=1, shuffle=False, num_workers=1)
    h = MyHandler(default_cfg(bcb_mode="abmil", bp_every_batch=4), device=DEV)
    load_synth(h.netG, "G-abmil:"); load_synth(h.netD, "D-prj:")
    h.patient_id.update({"train": [f"t{i}" for i in range(12)], "masked": [f"t{i}" for i in range(4)], "label_visible": [f"t{i}" for i in range(12)]})
    cache = ingest.device_bag_cache(DEV)
    hits0, miss0 = cache.hits, cache.misses
    evals = []
    for epoch in range(2):
        cl = h._train_each_epoch(train, "train", "wlabel")
        assert cl["y"].shape == (12, 2) and cl["y_hat"].shape == (12, 1) and bool(torch.isfinite(cl["y_hat"]).all())
        evals.append(MyHandler.test_model(h.netG, h.netD, "abmil", val, times_test_sample=1, test_zero_noise=True))
        assert evals[-1]["idx"].reshape(-1).tolist() == [0, 1, 2, 3, 4] and bool(torch.isfinite(evals[-1]["f_fake"]).all())
        if epoch == 0:
            assert cache.misses - miss0 == 17 and cache.hits == hits0                  # every bag came over PCIe once
    assert cache.misses - miss0 == 17 and cache.hits - hits0 == 17                       # second epoch: no host bag was staged
    assert len(h.pop_logs()) == 2 * 3 * 2
    view = h._bag_caches["train"]
    assert view.scope == ingest.dataset_scope(train) and view.stats()["bags"] == 12
    # same weights, zero noise: the evaluation of the validation set out of the cache equals its first evaluation over PCIe ... after
    # one more epoch of training the weights differ, so compare a third pass against the second instead
    again = MyHandler.test_model(h.netG, h.netD, "abmil", val, times_test_sample=1, test_zero_noise=True)
    assert torch.equal(again["y_hat"], evals[1]["y_hat"]) and torch.equal(again["f_fake"], evals[1]["f_fake"])
    per_bag = MyHandler.test_model(h.netG, h.netD, "abmil", val, times_test_sample=1, test_zero_noise=True, batch_bags=1)
    assert float((per_bag["y_hat"] - again["y_hat"]).abs().max()) <= 2e-6
    # another dataset with the same patient indices is another scope
    other = DataLoader(Patients(900, (512, 768, 640, 896)), batch_size=1, shuffle=False)
    m1 = cache.misses
    h._train_each_epoch(other, "train", "wlabel")
    assert cache.misses - m1 == 4
    # a randomly masked dataset (ratio_mask) is never kept
    masked = DataLoader(Patients(950, (512, 768, 640, 896), ratio_mask=0.2), batch_size=1, shuffle=False)
    n0 = len(cache.entries)
    h._train_each_epoch(masked, "masked", "wlabel")
    assert len(cache.entries) == n0 and h._bag_caches["masked"] is None
