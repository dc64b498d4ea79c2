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


def _lmdb_items(images, labels, corrupt=()):
    """the key scheme of the reference's tools/create_lmdb_dataset.py:327-345: num-samples, label-%09d, image-%09d (1-based)"""
    items = {b"num-samples": str(len(labels)).encode()}
    for i, (a, lab) in enumerate(zip(images, labels), 1):
        buf = io.BytesIO()
        PIL.Image.fromarray(a).save(buf, format="PNG")
        items[b"image-%09d" % i] = b"not an image" if i in corrupt else buf.getvalue()
        items[b"label-%09d" % i] = lab.encode("utf-8")
    return items


@pytest.mark.parametrize("page_size,leaf_fill", [(4096, 1.0), (4096, 0.35), (512, 0.4)])
def test_lmdb_leaf_through_the_mdb_walker(tmp_path, page_size, leaf_fill):
    """LmdbDataset (reference data/dataset.py:44-112) over a data.mdb read by mrn_amd/data/mdb.py (no `lmdb` package in this image):
    50 PNG crops + labels in the reference's key scheme, values inline and in overflow runs, trees of depth 2 and 3; every sample, the
    length filter, the corrupted-image path and the collated batch agree with the in-memory ArrayDataset of the same samples.
    Self-pinned: the file comes from tests/mdb_writer.py (no LMDB implementation is available to write it)."""
    from mrn_amd.data.dataset import AlignCollate, ArrayDataset, LmdbDataset, hierarchical_dataset
    from mrn_amd.data.mdb import Environment
    from tests.mdb_writer import write_environment
    opt = make_opt()
    images, labels = fake_text_samples("rootA/Latin")
    images, labels = images[:50], labels[:50]
    images = [a[:12, :30] if i % 5 == 0 else (a[:4, :6] if i % 5 == 1 else a) for i, a in enumerate(images)]   # ~1.5 KB / ~150 B PNGs stay inline
    labels[7] = "h\u00e9llo w\u00f6rld \u4e2d\u6587"                                          # multi-byte UTF-8 label
    leaf = tmp_path / "train" / "Latin"
    items = _lmdb_items(images, labels, corrupt=(3,))
    st = write_environment(str(leaf), items, page_size=page_size, leaf_fill=leaf_fill)
    env = Environment(str(leaf))
    assert env.stat() == dict(st, psize=page_size) and env.entries == 101
    assert st["overflow_pages"] > 0 and st["leaf_pages"] > 1 and st["depth"] >= 2                # big values AND inline values, a real tree
    if page_size == 512:
        assert st["depth"] >= 3
    assert list(env.items()) == sorted(items.items())
    assert env.get(b"image-000000000") is None and env.get(b"label-000000051") is None and env.get(b"zz") is None
    ds = LmdbDataset(str(leaf), opt)
    ref = ArrayDataset(images, labels, opt)
    keep = [i for i in range(50) if len(labels[i]) <= 25]
    assert len(ds) == len(ref) == len(keep)
    for j, i in enumerate(keep):
        img, lab = ds[j]
        rimg, rlab = ref[j]
        if i == 2:                                                                 # image-000000003 is corrupted
            assert lab == "[dummy_label]" and img.size == (256, 32)
        else:
            assert lab == rlab == labels[i] and img.mode == "RGBA" and np.array_equal(np.asarray(img), np.asarray(rimg))
    with contextlib.redirect_stdout(io.StringIO()):
        tree, log = hierarchical_dataset(str(tmp_path / "train"), opt, select_data="/")
    assert len(tree) == len(keep) and "Latin" in log
    sel = [j for j, i in enumerate(keep) if i != 2][:4]
    b1, l1 = AlignCollate(opt)([tree[j] for j in sel])
    b2, l2 = AlignCollate(opt)([ref[j] for j in sel])
    assert torch.equal(b1, b2) and list(l1) == list(l2) and tuple(b1.shape) == (4, 4, 32, 256)


def test_mdb_walker_rejects_what_it_does_not_understand(tmp_path):
    """a wrong magic / version, a truncated file, a page that carries another page's number: MdbError, never a guess"""
    from mrn_amd.data.mdb import Environment, MdbError
    from tests.mdb_writer import write_environment
    images, labels = fake_text_samples("rootA/Latin")
    write_environment(str(tmp_path / "ok"), _lmdb_items(images[:5], labels[:5]))
    good = (tmp_path / "ok" / "data.mdb").read_bytes()
    assert Environment(str(tmp_path / "ok")).get(b"num-samples") == b"5"

    def broken(name, data):
        d = tmp_path / name
        d.mkdir()
        (d / "data.mdb").write_bytes(data)
        return str(d)
    bad_magic = bytearray(good)
    bad_magic[16:20] = b"\0\0\0\0"
    with pytest.raises(MdbError):
        Environment(broken("magic", bytes(bad_magic)))
    bad_version = bytearray(good)
    bad_version[20:24] = (2).to_bytes(4, "little")
    with pytest.raises(MdbError):
        Environment(broken("version", bytes(bad_version)))
    with pytest.raises(MdbError):
        Environment(broken("short", good[:700]))
    swapped = bytearray(good)
    swapped[2 * 4096:2 * 4096 + 8] = (9).to_bytes(8, "little")                     # page 2 claims to be page 9
    env = Environment(broken("pgno", bytes(swapped)))
    with pytest.raises(MdbError):
        list(env.items())
    trunc = Environment(broken("trunc", good[:len(good) - 4096]))                 # the last page is missing
    with pytest.raises(MdbError):
        list(trunc.items())
    # an empty environment (no write transaction yet): every lookup misses
    write_environment(str(tmp_path / "empty"), {})
    e = Environment(str(tmp_path / "empty"))
    assert e.get(b"num-samples") is None and list(e.items()) == []


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
