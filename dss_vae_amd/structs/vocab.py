"""Vocabulary containers with the reference's attribute layout, so that a reference ``vocab.bin``
pickle resolves onto these classes (dss_vae/structs/vocab.py:10-93)."""
import pickle


class VocabEntry(object):
    """word <-> id map; <pad>=0, <s>=1, </s>=2, <unk>=3; unknown words map to <unk>."""

    def __init__(self):
        self.word2id = {"<pad>": 0, "<s>": 1, "</s>": 2, "<unk>": 3}
        self.pad_id, self.sos_id, self.eos_id, self.unk_id = 0, 1, 2, 3
        self.id2word = {i: w for w, i in self.word2id.items()}

    def __getitem__(self, word):
        return self.word2id.get(word, self.unk_id)

    def __contains__(self, word):
        return word in self.word2id

    def __setitem__(self, key, value):
        raise ValueError("vocabulary is readonly")

    def __len__(self):
        return len(self.word2id)

    def __repr__(self):
        return "Vocabulary[size=%d]" % len(self)

    def add(self, word):
        if word in self.word2id:
            return self.word2id[word]
        wid = self.word2id[word] = len(self.word2id)
        self.id2word[wid] = word
        return wid

    def is_unk(self, word):
        return word not in self.word2id

    @staticmethod
    def from_corpus(corpus, size=50000, freq_cutoff=0):
        from collections import Counter
        freq = Counter(w for sent in corpus for w in sent)
        entry = VocabEntry()
        for word in sorted(freq, key=freq.get, reverse=True)[:size]:
            if len(entry) < size and freq[word] >= freq_cutoff:
                entry.add(word)
        return entry


class Vocab(object):
    def __init__(self, **kwargs):
        self.entries = []
        for key, item in kwargs.items():
            assert isinstance(item, VocabEntry)
            setattr(self, key, item)
            self.entries.append(key)

    def __repr__(self):
        return "Vocab(%s)" % ", ".join("%s %swords" % (e, getattr(self, e)) for e in self.entries)

    @staticmethod
    def from_bin_file(file_path):
        import os
        if os.path.getsize(file_path) <= 0:
            raise RuntimeWarning("Empty vocab")
        from .io import load_bin
        return load_bin(file_path)

    def save(self, file_path):
        from .io import dump_bin
        dump_bin(self, file_path)
