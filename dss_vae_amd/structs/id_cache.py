"""Dataset-level id cache ("a one-time conversion to a flat id/offset tensor cache").

The reference re-tensorises every minibatch in Python on every forward (structs/dataset.py:64-75 `batch_iter` ->
utils/nn_funcs.py:143-174 `to_input_variable` -> :43-47 `word2id`, a dict lookup per token, three times per step).
Here `train.bin` is converted ONCE into four flat arrays -- the source ids of all sentences back to back with their
offsets, and the same for the linearised trees -- which live in HBM for the whole run (100 k sentences x (30 + 90)
tokens x 8 B = 96 MB of 288 GB).  A minibatch is then a list of example indices: the host shuffles and length-sorts
indices with numpy (no token is touched), ships B int64 indices, and the padded `src` / `<s> src </s>` / `<s> tree </s>`
matrices, the length vectors and the non-pad row lists are cut on the device (ops.batch_rows / ops.keep_rows).
`batch_iter` yields the same batches as `Dataset.batch_iter` under the same numpy RNG (same shuffle, same stable
length sort), and the same shards under data parallel as parallel.shard_examples."""
import os

import numpy as np
import torch

from .. import ops
from ..vae.base_vae import DeviceBatch, Lengths


def _flatten(seqs, entry):
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    np.cumsum([len(s) for s in seqs], out=off[1:])
    w2i, unk = entry.word2id, entry.unk_id
    flat = np.fromiter((w2i.get(w, unk) for s in seqs for w in s), dtype=np.int64, count=int(off[-1]))
    return flat, off


class IdCache(object):
    def __init__(self, src_ids, src_off, tgt_ids, tgt_off, vocab):
        self.src_ids, self.src_off, self.tgt_ids, self.tgt_off, self.vocab = src_ids, src_off, tgt_ids, tgt_off, vocab
        self.src_len = np.diff(src_off)
        self.tgt_len = None if tgt_off is None else np.diff(tgt_off)
        self.dev = None

    def __len__(self):
        return len(self.src_len)

    # ---- one-time conversion ---------------------------------------------------------------------
    @staticmethod
    def from_examples(examples, vocab):
        """examples: PlainExample list (`src` / `tgt` token lists; tgt None everywhere = no trees)"""
        src_ids, src_off = _flatten([e.src for e in examples], vocab.src)
        tgt_ids = tgt_off = None
        if len(examples) and examples[0].tgt is not None:
            tgt_ids, tgt_off = _flatten([e.tgt for e in examples], vocab.tgt)
        return IdCache(src_ids, src_off, tgt_ids, tgt_off, vocab)

    def save(self, path):
        arrays = dict(src_ids=self.src_ids, src_off=self.src_off, n_src=np.int64(len(self.vocab.src)),
                      n_tgt=np.int64(len(self.vocab.tgt)))
        if self.tgt_ids is not None:
            arrays.update(tgt_ids=self.tgt_ids, tgt_off=self.tgt_off)
        tmp = path + ".tmp%d.npz" % os.getpid()
        np.savez(tmp, **arrays)
        os.replace(tmp, path)

    @staticmethod
    def load(path, vocab):
        """-> IdCache, or None when the file was written for other vocabularies"""
        with np.load(path, allow_pickle=False) as z:
            if int(z["n_src"]) != len(vocab.src) or int(z["n_tgt"]) != len(vocab.tgt):
                return None
            has_tgt = "tgt_ids" in z.files
            return IdCache(z["src_ids"], z["src_off"], z["tgt_ids"] if has_tgt else None,
                           z["tgt_off"] if has_tgt else None, vocab)

    @staticmethod
    def for_file(bin_path, examples, vocab, write=True):
        """the cache of a `train.bin`: read from `<bin_path>.ids.npz` when that file is newer than the pickle and was
        made for these vocabularies, else converted now (and written there when `write`)"""
        path = bin_path + ".ids.npz"
        try:
            if os.path.getmtime(path) >= os.path.getmtime(bin_path):
                hit = IdCache.load(path, vocab)
                if hit is not None and len(hit) == len(examples):
                    return hit
        except (OSError, KeyError, ValueError):
            pass
        cache = IdCache.from_examples(examples, vocab)
        if write:
            try:
                cache.save(path)
            except OSError:
                pass
        return cache

    # ---- device side -----------------------------------------------------------------------------
    def cuda(self, device=None):
        if self.dev is None:
            from ..utils.nn_funcs import device as current
            d = current() if device is None else device
            up = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(d)
            self.dev = dict(src_ids=up(self.src_ids), src_off=up(self.src_off), tgt_ids=up(self.tgt_ids),
                            tgt_off=up(self.tgt_off), device=d)
        return self

    def batch(self, idx, with_tgt=True):
        """idx: numpy int64 example indices, already in the batch's order (descending source length) -> DeviceBatch
        whose tensors are cut on the device; the host only looks at the cached length arrays"""
        self.cuda()
        d = self.dev
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        idx_dev = torch.from_numpy(idx).pin_memory().to(d["device"], non_blocking=True)
        B = len(idx)
        sl = self.src_len[idx]
        L = int(sl.max())
        length = Lengths(sl.tolist())
        length.dev = torch.empty(B, device=d["device"], dtype=torch.int32)
        v = self.vocab
        src = ops.batch_rows(d["src_ids"], d["src_off"], idx_dev, L, False, v.src, lengths_out=length.dev)
        tgt_var = ops.batch_rows(d["src_ids"], d["src_off"], idx_dev, L + 2, True, v.src)
        syn_var = tree_length = None
        if with_tgt and d["tgt_ids"] is not None:
            tl = self.tgt_len[idx]
            tree_length = Lengths(tl.tolist())
            tree_length.dev = torch.empty(B, device=d["device"], dtype=torch.int32)
            syn_var = ops.batch_rows(d["tgt_ids"], d["tgt_off"], idx_dev, int(tl.max()) + 2, True, v.tgt,
                                     lengths_out=tree_length.dev)
        return DeviceBatch.from_ids(src, length, tgt_var, syn_var, tree_length)

    def batch_indices(self, batch_size, shuffle=False, rng=None, rank=0, world=1):
        """the index lists of `Dataset.batch_iter(batch_size * world, shuffle, rng)` dealt to `rank` of `world` as
        parallel.shard_examples does: -> (global batch size, this rank's indices sorted by descending length)"""
        order = np.arange(len(self))
        if shuffle:
            (rng or np.random).shuffle(order)
        step = batch_size * world
        for lo in range(0, len(order), step):
            cut = order[lo:lo + step]
            cut = cut[np.argsort(-self.src_len[cut], kind="stable")]  # list.sort(key=-len) is stable too
            yield len(cut), (cut[rank::world] if world > 1 else cut)

    def batch_iter(self, batch_size, shuffle=False, rng=None, rank=0, world=1, with_tgt=True):
        """-> (global batch size, DeviceBatch or None for an empty shard) per step.  Cutting a batch is asynchronous
        (one small pinned host-to-device copy and a few microsecond-sized launches, all stream-ordered), so the host
        cuts step k + 1's batch while the GPU is still busy with step k: no separate prefetch stage is needed."""
        for n_global, idx in self.batch_indices(batch_size, shuffle, rng, rank, world):
            yield n_global, (self.batch(idx, with_tgt=with_tgt) if len(idx) else None)
