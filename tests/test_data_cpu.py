"""The input pipeline (mrn_amd/data/{dataset,data_manage}.py) against the reference's own classes run on in-memory datasets
(tests/golden/data_manage.npz from tests/golden/make_golden_il.py): which loaders get_dataset() builds for every `memory`
mode, the batches they yield under fixed numpy / torch seeds (labels, rehearsal / current domain indices, resized and
normalised pixels), the validation loaders, ResizeNormalize."""
import contextlib
import io
import types

import numpy as np
import PIL.Image
import pytest
import torch

from tests.helpers import fake_text_samples, load_golden


def make_opt(**kw):
    o = types.SimpleNamespace(imgH=32, imgW=256, batch_max_length=25, memory_num=40, batch_size=6, workers=0, Aug="None", il="mrn",
                              memory="random", lan_list=["Chinese", "Latin", "Japanese"], select_data=["rootA", "rootB"],
                              device_prefetch=False)
    o.__dict__.update(kw)
    return o


def open_fake(path, opt, mode="train"):
    from mrn_amd.data.dataset import ArrayDataset
    images, labels = fake_text_samples(path)
    return ArrayDataset(images, labels, opt, mode)


def open_fake_tree(root, opt, mode="train"):
    from torch.utils.data import ConcatDataset
    return ConcatDataset([open_fake(root, opt, mode)]), "log"


SCENARIOS = [("mrn_random", "mrn", "random"), ("plain", "mrn", None), ("test_ch", "lwf", "test_ch"), ("large", "lwf", "large"),
             ("total", "lwf", "total"), ("halves", "lwf", "random")]


@pytest.mark.parametrize("name,il,memory", SCENARIOS)
def test_dataset_manager_matches_reference(name, il, memory):
    from mrn_amd.data.data_manage import Dataset_Manager
    from mrn_amd.data.dataset import AlignCollate2
    g = load_golden("data_manage")
    opt = make_opt(il=il, memory=memory, memory_num=40 if memory != "large" else 12)
    np.random.seed(77)
    torch.manual_seed(77)
    with contextlib.redirect_stdout(io.StringIO()):
        dm = Dataset_Manager(opt, open_dataset=open_fake, device=torch.device("cpu"))
        dm.select_data = opt.select_data
        taski = 2
        index_list = [np.random.choice(range(30), 40 // taski if memory != "large" else 12, replace=False) for _ in range(taski)]
        dm.get_dataset(taski, memory=memory, index_list=index_list)
        assert len(dm.data_loader_list) == int(g[f"{name}/n_loaders"])
        assert [len(l.dataset) for l in dm.data_loader_list] == list(g[f"{name}/loader_dataset_lengths"])
        assert [l.batch_size for l in dm.data_loader_list] == list(g[f"{name}/loader_batch_sizes"])
        mix = isinstance(dm.data_loader_list[0].collate_fn, AlignCollate2)
        assert mix == bool(g[f"{name}/mix"])
        for b in range(3):
            got = dm.get_batch2() if mix else dm.get_batch()
            assert list(got[1]) == [str(s) for s in g[f"{name}/batch{b}/labels"]]
            assert list(got[0].shape) == list(g[f"{name}/batch{b}/image_shape"])
            assert np.array_equal(got[0][:, :, ::8, ::32].numpy(), g[f"{name}/batch{b}/image_probe"])      # bit-exact pixels
            assert abs(got[0].double().sum().item() - float(g[f"{name}/batch{b}/image_sum"])) < 1e-6
            if mix:
                assert np.array_equal(np.array([list(t) for t in got[2]]), g[f"{name}/batch{b}/index"])
        _, n = dm.rehearsal_prev_model(taski)
        assert n == int(g[f"{name}/prev_len"])


def test_validation_loaders_and_resize_normalize_match_reference():
    from mrn_amd.data.data_manage import Val_Dataset
    from mrn_amd.data.dataset import ResizeNormalize
    g = load_golden("data_manage")
    opt = make_opt(batch_size=5, lan_list=["Chinese", "Latin"])
    np.random.seed(78)
    torch.manual_seed(78)
    with contextlib.redirect_stdout(io.StringIO()):
        vd = Val_Dataset(["valA/Chinese", "valA/Latin"], opt, open_tree=open_fake_tree)
        for name, loader in (("val/current", vd.create_dataset()), ("val/list", vd.create_list_dataset())):
            assert len(loader.dataset) == int(g[f"{name}/len"])
            images, labels = next(iter(loader))
            assert list(labels) == [str(s) for s in g[f"{name}/labels"]]
            assert abs(images.double().sum().item() - float(g[f"{name}/image_sum"])) < 1e-6
    img, _ = open_fake("rootA/Latin", opt)[3]
    t = ResizeNormalize((256, 32))(img)
    assert list(t.shape) == list(g["resize/shape"]) and t.dtype == torch.float32
    assert np.array_equal(t[:, ::4, ::16].numpy(), g["resize/probe"])
    assert float(t.min()) >= -1.0 and float(t.max()) <= 1.0


def test_npz_leaf_round_trip_and_label_filter(tmp_path):
    """<dir>/data.npz is the portable sibling of an LMDB leaf: same samples, same length filter, same corrupted-image handling"""
    from mrn_amd.data.dataset import AlignCollate, NpzDataset, hierarchical_dataset
    opt = make_opt()
    images, labels = fake_text_samples("rootA/Latin")
    enc = []
    for a in images[:6]:
        buf = io.BytesIO()
        PIL.Image.fromarray(a).save(buf, format="PNG")
        enc.append(np.frombuffer(buf.getvalue(), dtype=np.uint8))
    enc[2] = np.frombuffer(b"not an image", dtype=np.uint8)                       # corrupted sample
    leaf = tmp_path / "train" / "Latin"
    leaf.mkdir(parents=True)
    obj = np.empty(6, dtype=object)
    for i, e in enumerate(enc):
        obj[i] = e
    np.savez(leaf / "data.npz", images=obj, labels=np.array(labels[:6]))
    ds = NpzDataset(str(leaf), opt)
    keep = [i for i in range(6) if len(labels[i]) <= 25]
    assert len(ds) == len(keep)
    k2 = keep.index(2) if 2 in keep else None
    for j, i in enumerate(keep):
        img, lab = ds[j]
        assert img.mode == "RGBA"
        if j == k2:
            assert lab == "[dummy_label]" and img.size == (256, 32)
        else:
            assert lab == labels[i] and np.array_equal(np.asarray(img), images[i])        # PNG is lossless
    with contextlib.redirect_stdout(io.StringIO()):
        tree, log = hierarchical_dataset(str(tmp_path / "train"), opt, select_data="/")
    assert len(tree) == len(keep) and "Latin" in log
    batch, labs = AlignCollate(opt)([tree[0], tree[1]])
    assert tuple(batch.shape) == (2, 4, 32, 256)
    with pytest.raises(ImportError):
        from mrn_amd.data.dataset import LmdbDataset
        LmdbDataset(str(leaf), opt)                                                # lmdb is not installed in this image


def test_ranks_draw_different_shards_of_the_global_batch():
    """data parallelism (mrn_amd/tiny_train.py with N > 1 ranks): every rank's Dataset_Manager yields batch_size // world samples
    per loader from its own shuffle order -- the ranks' batches differ, the rehearsal-memory subsets (numpy seed) agree, and
    world == 1 keeps the full batch"""
    from mrn_amd.data.data_manage import Dataset_Manager
    opt = make_opt(il="mrn", memory="random", memory_num=40, batch_size=8, manual_seed=111)
    batches, loaders = [], []
    for rank in (0, 1):
        np.random.seed(77)
        torch.manual_seed(77)
        with contextlib.redirect_stdout(io.StringIO()):
            dm = Dataset_Manager(opt, open_dataset=open_fake, device=torch.device("cpu"), rank=rank, world=2)
            dm.select_data = opt.select_data
            index_list = [np.random.choice(range(30), 20, replace=False) for _ in range(2)]
            dm.get_dataset(2, memory="random", index_list=index_list)
            batches.append([dm.get_batch2() for _ in range(2)])
            loaders.append(dm.data_loader_list)
    assert [l.batch_size for l in loaders[0]] == [l.batch_size for l in loaders[1]] == [4] * len(loaders[0])
    assert [len(l.dataset) for l in loaders[0]] == [len(l.dataset) for l in loaders[1]]          # same rehearsal subsets
    (img0, lab0, _), (img1, lab1, _) = batches[0][0], batches[1][0]
    assert img0.shape == img1.shape and img0.shape[0] == 4 * len(loaders[0])
    assert list(lab0) != list(lab1) and not torch.equal(img0, img1), "both ranks drew the same first batch"
    with contextlib.redirect_stdout(io.StringIO()):
        dm = Dataset_Manager(opt, open_dataset=open_fake, device=torch.device("cpu"))
        dm.select_data = opt.select_data
        dm.get_dataset(2, memory="random", index_list=index_list)
    assert [l.batch_size for l in dm.data_loader_list] == [8] * len(dm.data_loader_list)
