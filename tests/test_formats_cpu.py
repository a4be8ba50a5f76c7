"""Host-side rows that need no GPU: on-disk formats, corpus BLEU (N3), loss tracker."""
import os
import pickle
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from dss_vae_amd.metrics.bleu import corpus_bleu, evaluate_file  # noqa: E402
from dss_vae_amd.structs import Dataset, PlainExample, Vocab, VocabEntry  # noqa: E402
from dss_vae_amd.structs.io import dump_bin, load_bin, load_model_bin  # noqa: E402


def _vocab():
    v = Vocab(src=VocabEntry(), tgt=VocabEntry())
    for w in ("hello", "world"):
        v.src.add(w)
    v.tgt.add("NP")
    return v


def test_bin_files_use_the_reference_class_paths_and_round_trip(tmp_path):
    v = _vocab()
    v.save(str(tmp_path / "vocab.bin"))
    raw = (tmp_path / "vocab.bin").read_bytes()
    assert b"dss_vae.structs.vocab" in raw and b"dss_vae_amd" not in raw
    v2 = Vocab.from_bin_file(str(tmp_path / "vocab.bin"))
    assert type(v2) is Vocab and type(v2.src) is VocabEntry
    assert v2.src["hello"] == 4 and v2.src["never-seen"] == v2.src.unk_id and v2.entries == ["src", "tgt"]
    assert (v2.src.pad_id, v2.src.sos_id, v2.src.eos_id, v2.src.unk_id) == (0, 1, 2, 3)
    ds = Dataset([PlainExample(["a", "b", "c"], ["NP", "/BRK"]), PlainExample(["d"], ["NP"])])
    ds.save(str(tmp_path / "train.bin"))
    assert isinstance(pickle_load_plain(tmp_path / "train.bin"), list)
    back = Dataset.from_bin_file(str(tmp_path / "train.bin"))
    assert [e.src for e in back.examples] == [["a", "b", "c"], ["d"]] and back.examples[0].tgt == ["NP", "/BRK"]
    batches = list(back.batch_iter(2, shuffle=False))
    assert [len(e.src) for e in batches[0]] == [3, 1]
    assert Vocab.__module__ == "dss_vae_amd.structs.vocab"


def pickle_load_plain(path):
    """the file is an ordinary pickle whose globals live under dss_vae.structs.*: resolve them by hand"""
    from dss_vae_amd.structs.io import _Unpickler
    with open(path, "rb") as f:
        return _Unpickler(f).load()


def test_unpickler_refuses_foreign_globals(tmp_path):
    class Evil(object):
        def __reduce__(self):
            return (os.system, ("true",))

    (tmp_path / "evil.bin").write_bytes(pickle.dumps(Evil()))
    with pytest.raises(pickle.UnpicklingError):
        load_bin(str(tmp_path / "evil.bin"))


@pytest.mark.skipif(not os.path.isdir("/root/reference/dss_vae"), reason="reference checkout not present")
def test_interchange_with_the_reference_classes(tmp_path):
    """files written here are read by the reference's own loaders and vice versa (build container only)"""
    import subprocess
    _vocab().save(str(tmp_path / "vocab.bin"))
    Dataset([PlainExample(["a", "b"], ["t"])]).save(str(tmp_path / "train.bin"))
    code = r'''
import sys, types, pickle
sys.path.insert(0, "/root/reference")
for n in ("nltk", "nltk.tree"):
    m = types.ModuleType(n); m.Tree = object; sys.modules.setdefault(n, m)
from dss_vae.structs.vocab import Vocab
from dss_vae.structs.dataset import Dataset
v = Vocab.from_bin_file(sys.argv[1] + "/vocab.bin")
assert type(v).__module__ == "dss_vae.structs.vocab" and v.src["hello"] == 4
d = Dataset.from_bin_file(sys.argv[1] + "/train.bin")
assert d.examples[0].src == ["a", "b"]
pickle.dump(v, open(sys.argv[1] + "/ref_vocab.bin", "wb"))
pickle.dump(d.examples, open(sys.argv[1] + "/ref_train.bin", "wb"))
'''
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    subprocess.run([sys.executable, "-c", code, str(tmp_path)], check=True, env=env)
    v = load_bin(str(tmp_path / "ref_vocab.bin"))
    assert type(v) is Vocab and v.src["world"] == 5
    assert Dataset.from_bin_file(str(tmp_path / "ref_train.bin")).examples[0].tgt == ["t"]


def test_model_bin_loads_without_executing_anything(tmp_path):
    from argparse import Namespace
    path = str(tmp_path / "model.bin")
    torch.save({"args": Namespace(a=1), "vocab": _vocab(), "state_dict": {"w": torch.ones(2, 3)}}, path)
    d = load_model_bin(path)
    assert d["args"].a == 1 and d["vocab"].src["hello"] == 4 and torch.equal(d["state_dict"]["w"], torch.ones(2, 3))


def test_corpus_bleu_known_answers(tmp_path):
    ref = "the cat sat on the mat".split()
    assert corpus_bleu([[ref]], [ref]) == pytest.approx(1.0)
    # every n-gram of the hypothesis matches, hypothesis one token short: BLEU = brevity penalty exp(1 - 6/5)
    assert corpus_bleu([[ref]], [ref[:-1]]) == pytest.approx(2.718281828459045 ** (1 - 6 / 5))
    assert corpus_bleu([[ref]], ["a b c d e f".split()]) == 0.0
    # no 4-gram match -> 0 without smoothing
    assert corpus_bleu([[ref]], ["the cat sat x the mat".split()]) == 0.0
    # pooled over the corpus: p1 = 9/10, p2 = 6/8, p3 = 4/6, p4 = 3/4, lengths equal
    hyps = [ref, "the dog sat on".split()]
    refs = [[ref], ["the cat sat on".split()]]
    want = (9 / 10 * 6 / 8 * 4 / 6 * 3 / 4) ** 0.25
    # second pair: 1-grams 3/4, 2-grams 1/3 ("sat on"), 3-grams 0/2, 4-grams 0/1
    assert corpus_bleu(refs, hyps) == pytest.approx(want)
    (tmp_path / "p.txt").write_text(" ".join(ref) + "\n")
    (tmp_path / "g.txt").write_text(" ".join(ref) + "\n")
    assert evaluate_file(str(tmp_path / "p.txt"), str(tmp_path / "g.txt")) == pytest.approx(100.0)


def test_update_tracker_concatenates_batch_means():
    from dss_vae_amd.utils.model_utils import update_tracker
    t = update_tracker({"Loss": torch.tensor([1.0, 3.0]), "KL Weight": 0.5}, {})
    t = update_tracker({"Loss": torch.tensor(4.0), "KL Weight": 0.25}, t)
    assert torch.equal(t["Loss"], torch.tensor([2.0, 4.0])) and t["KL Weight"] == 0.75
