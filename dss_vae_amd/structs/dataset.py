"""Example / dataset containers in the reference's pickle layout (dss_vae/structs/dataset.py:37-122):
``train.bin`` is a pickled *list* of PlainExample(src, tgt, idx, meta, entries)."""
import pickle

import numpy as np


class PlainExample(object):
    def __init__(self, src, tgt, idx=0, meta=None, **kwargs):
        self.entries = []
        for key, item in kwargs.items():
            setattr(self, key, item)
            self.entries.append(key)
        self.src, self.tgt, self.idx, self.meta = src, tgt, idx, meta

    @staticmethod
    def parse_raw(raw_line, field_split="\t", token_split=" "):
        items = raw_line.strip().split(field_split)
        tgt = items[1].split(token_split) if len(items) > 1 else None
        return PlainExample(src=items[0].split(token_split), tgt=tgt)

    @staticmethod
    def parse_for_control(raw_line, field_split="\t", token_split=" "):
        """`sentence \t reference` line of a controlled-generation file -> two source-only examples
        (structs/dataset.py:114-119)"""
        items = raw_line.strip().split(field_split)
        return (PlainExample(src=items[0].split(token_split), tgt=None),
                PlainExample(src=items[1].split(token_split), tgt=None))

    def __len__(self):
        return len(self.src)


def to_example(words, token_split=" "):
    return PlainExample(src=words.split(token_split), tgt=None)


class Dataset(object):
    def __init__(self, examples):
        self.examples = examples

    @staticmethod
    def from_bin_file(file_path):
        from .io import load_bin
        return Dataset(load_bin(file_path))

    def save(self, file_path):
        """a pickled LIST of examples, as structs/generate_dataset.py:146-157 writes it"""
        from .io import dump_bin
        dump_bin(list(self.examples), file_path)

    @staticmethod
    def from_raw_file(file_path, e_type="Plain"):
        """one `src [\t tgt]` line per example (structs/dataset.py:46-57).  The treebank example types ("PTB",
        "SyntaxGEN") parse bracketed trees with nltk and belong to preprocessing, which is out of scope."""
        if e_type not in ("Plain", "NAG", "SyntaxVAE"):
            raise NotImplementedError("example type %r needs the treebank preprocessing (nltk); only the plain "
                                      "`src \\t tgt` line format is read here" % (e_type,))
        with open(file_path, "r") as f:
            return Dataset([PlainExample.parse_raw(line) for line in f])

    @staticmethod
    def from_list(raw_list):
        return Dataset([PlainExample.parse_raw(line) for line in raw_list])

    def batch_iter(self, batch_size, shuffle=False, rng=None):
        """shuffle, slice, sort each batch by -len(src) (packing needs descending lengths).
        rng: a numpy RandomState for the shuffle (data parallel: the same on every rank); default: the
        global numpy state, as the reference."""
        order = np.arange(len(self.examples))
        if shuffle:
            (rng or np.random).shuffle(order)
        for lo in range(0, len(order), batch_size):
            batch = [self.examples[i] for i in order[lo:lo + batch_size]]
            batch.sort(key=lambda e: -len(e))
            yield batch

    def __len__(self):
        return len(self.examples)

    def __iter__(self):
        return iter(self.examples)
