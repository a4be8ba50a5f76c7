"""Dataset id cache (structs/id_cache.py), host side: the one-time conversion, its file, and the batch schedule --
the same index lists as `Dataset.batch_iter` (structs/dataset.py:64-75: shuffle, slice, stable sort by -len(src))
under the same numpy RNG, dealt to data-parallel ranks as parallel.shard_examples deals examples.  The device side
(batches cut by dssvae_batch_rows / dssvae_keep_rows) is tested on the GPU: tests/test_graph_gpu.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from parity_util import make_examples, product_examples, product_vocab  # noqa: E402


def _data(n=57, seed=3):
    vocab = product_vocab(200, 30)
    exs = product_examples(make_examples(n, 200, 30, src_len=12, tree_len=36, fixed=False, seed=seed))
    np.random.RandomState(seed).shuffle(exs)
    exs[4].src[1] = "out-of-vocabulary"
    return exs, vocab


def test_conversion_is_the_vocabulary_lookup():
    from dss_vae_amd.structs.id_cache import IdCache
    exs, vocab = _data()
    c = IdCache.from_examples(exs, vocab)
    assert len(c) == len(exs) and c.src_off[-1] == sum(len(e.src) for e in exs)
    for i in (0, 4, 17, len(exs) - 1):
        assert c.src_ids[c.src_off[i]:c.src_off[i + 1]].tolist() == [vocab.src[w] for w in exs[i].src]
        assert c.tgt_ids[c.tgt_off[i]:c.tgt_off[i + 1]].tolist() == [vocab.tgt[w] for w in exs[i].tgt]
    assert c.src_ids[c.src_off[4] + 1] == vocab.src.unk_id
    assert c.src_len.tolist() == [len(e.src) for e in exs] and c.tgt_len.tolist() == [len(e.tgt) for e in exs]
    # sentences without trees (dev / test files of plain text)
    for e in exs:
        e.tgt = None
    c2 = IdCache.from_examples(exs, vocab)
    assert c2.tgt_ids is None and c2.tgt_len is None and np.array_equal(c2.src_ids, c.src_ids)


def test_schedule_equals_dataset_batch_iter():
    from dss_vae_amd.parallel import shard_examples
    from dss_vae_amd.structs import Dataset
    from dss_vae_amd.structs.id_cache import IdCache
    exs, vocab = _data()
    c = IdCache.from_examples(exs, vocab)
    pos = {id(e): i for i, e in enumerate(exs)}
    for world in (1, 2, 3):
        for shuffle in (False, True):
            ref = list(Dataset(exs).batch_iter(8 * world, shuffle=shuffle, rng=np.random.RandomState(5)))
            for rank in range(world):
                got = list(c.batch_indices(8, shuffle=shuffle, rng=np.random.RandomState(5), rank=rank, world=world))
                assert len(got) == len(ref)
                for batch, (n_global, idx) in zip(ref, got):
                    assert n_global == len(batch)
                    want = shard_examples(batch, rank, world) if world > 1 else batch
                    assert [pos[id(e)] for e in want] == idx.tolist()


def test_cache_file_round_trip_and_invalidation(tmp_path):
    from dss_vae_amd.structs import Dataset
    from dss_vae_amd.structs.id_cache import IdCache
    exs, vocab = _data()
    path = str(tmp_path / "train.bin")
    Dataset(exs).save(path)
    c = IdCache.for_file(path, exs, vocab)
    assert os.path.exists(path + ".ids.npz")
    hit = IdCache.for_file(path, exs, vocab, write=False)
    for name in ("src_ids", "src_off", "tgt_ids", "tgt_off"):
        assert np.array_equal(getattr(hit, name), getattr(c, name)), name
    # a cache written for other vocabularies is not used
    other = product_vocab(150, 30)
    assert IdCache.load(path + ".ids.npz", other) is None
    redo = IdCache.for_file(path, exs, other, write=False)
    assert redo.src_ids.max() < 150
    # ... nor one older than the pickle, nor one of another length
    os.utime(path + ".ids.npz", (1, 1))
    again = IdCache.for_file(path, exs[:10], vocab, write=False)
    assert len(again) == 10
