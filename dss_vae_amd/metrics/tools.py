"""Small host-side helpers of the evaluators (reference: dss_vae/metrics/tools.py:17-164): result files, the
nested list shapes the decode path hands around, example splitting, BLEU over decode results.

Shapes: a decode result is `[[tokens], [length]]`; `predict` returns a list of them; a reference is
`[[tokens]]` (one gold sentence per example)."""
from collections import Counter
from itertools import chain

import numpy as np

from ..structs import Dataset, PlainExample
from ..utils.nn_funcs import id2word, word2id
from . import bleu
from .evaluation import batchify_examples, predict_to_plain, prepare_ref, write_result  # noqa: F401  (same module in the reference)


def write_append_result(docs, fname):
    """write_result in append mode (tools.py:35-50)"""
    import torch
    with open(fname, "a") as f:
        if isinstance(docs, list):
            for doc in docs:
                f.write(" ".join(doc) if isinstance(doc, list) else doc)
                f.write("\n")
        elif isinstance(docs, dict):
            for key, val in docs.items():
                if val is not None:
                    f.write("%s:%f" % (key, torch.mean(val) if isinstance(val, torch.Tensor) else val))
                    f.write("\n")


class TagJudge(object):
    """is a token of the linearised tree a bracket tag rather than a word (tools.py:53-60)"""

    def __init__(self, vocab):
        self.vocab, self.unk = vocab, vocab.unk_id

    def is_tag(self, item):
        return self.vocab[item] > self.unk


def to_plain(result):
    return [r[0][0] for r in result]


def con_to_plain(decode_results):
    return [" ".join(hyp[0]) for hyp in decode_results]


def reference_to_plain(decode_results):
    return [" ".join(hyp[0][0][0]) for hyp in decode_results]


def recovery(seqs, vocab, keep_origin=False):
    """tokens as the model sees them (out-of-vocabulary words -> <unk>) unless keep_origin"""
    if keep_origin:
        return [seqs]
    return id2word(word2id(seqs, vocab), vocab)


def recovery_ref(refs, vocab, keep_origin=False):
    if keep_origin:
        return [[e.src] for e in refs]
    return [[recovery(e.src, vocab)] for e in refs]


def get_bleu_score(references, hypothesis):
    """references: one list of gold token lists per example; hypothesis: decode results -> corpus BLEU x 100"""
    return 100.0 * bleu.corpus_bleu([list(r) for r in references], [hyp[0][0] for hyp in hypothesis])


def prepare_input(examples, sort_key="src", batch_size=None):
    """batches in corpus order, each sorted by -len (tools.py:105-113)"""
    out = []
    for batch in Dataset(examples).batch_iter(50 if batch_size is None else batch_size, shuffle=False):
        batch.sort(key=lambda e: -len(getattr(e, sort_key)))
        out.append(batch)
    return out


def split_examples(examples):
    """`a \\t b` pair examples -> (examples of the a side, examples of the b side), source-only (tools.py:156-164)"""
    return ([PlainExample(src=e.src, tgt=None) for e in examples],
            [PlainExample(src=e.tgt, tgt=None) for e in examples])


# ---- corpus statistics of generated text (metrics/unigram_kl.py, structs/corpus.py:7-20) ------------
def avg_length(sentences):
    return sum(len(s) for s in sentences) / float(len(sentences))


def unigram_distribution(sentences, vocab):
    """relative unigram frequencies over the vocabulary, out-of-vocabulary words counted as <unk>, unseen words
    smoothed with 1 / N.  (The reference's CorpusDistribution reads `word_freq['word']` -- the count of the literal
    token "word" -- for every entry, so its figure is not a unigram statistic; this is the formula its docstring
    describes.  Parity unpinned.)"""
    counts = np.zeros(len(vocab), dtype=np.float64)
    for w, c in Counter(chain(*sentences)).items():
        counts[vocab[w]] += c
    total = counts.sum()
    counts = counts + (counts == 0) * (1.0 / max(total, 1.0))
    return counts / counts.sum()


def unigram_divergence(corpus_sentences, generated_sentences, vocab, dtype="js"):
    p, q = unigram_distribution(corpus_sentences, vocab), unigram_distribution(generated_sentences, vocab)

    def kl(a, b):
        return float(np.sum(a * (np.log(a) - np.log(b))))
    if dtype == "kl":
        return kl(p, q)
    m = 0.5 * (p + q)
    return 0.5 * kl(p, m) + 0.5 * kl(q, m)

