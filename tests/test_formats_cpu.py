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
    from argparse import Namespace
    from dss_vae_amd.structs.io import save_model_bin
    save_model_bin({"args": Namespace(a=1), "vocab": _vocab(), "state_dict": {"w": torch.ones(2, 3)}},
                   str(tmp_path / "model.bin"))
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
# model.bin as BaseVAE.save writes it (a file of this package's own making): the reference's torch.load
# resolves the vocabulary onto ITS classes
import torch
ck = torch.load(sys.argv[1] + "/model.bin", map_location="cpu", weights_only=False)
assert type(ck["vocab"]).__module__ == "dss_vae.structs.vocab" and ck["vocab"].src["hello"] == 4
assert ck["args"].a == 1 and float(ck["state_dict"]["w"].sum()) == 6.0
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
    # ... and as BaseVAE.save writes it: vocabulary under the reference's class paths
    from dss_vae_amd.structs.io import save_model_bin
    save_model_bin({"args": Namespace(a=2), "vocab": _vocab(), "state_dict": {"w": torch.ones(2, 3)}}, path)
    d = load_model_bin(path)
    assert d["args"].a == 2 and type(d["vocab"]) is Vocab and d["vocab"].src["hello"] == 4
    assert type(_vocab()).__module__ == "dss_vae_amd.structs.vocab"  # the presentation was undone


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


def test_raw_file_control_lines_padding_and_bow_helper(tmp_path):
    """host-side helpers with the reference's names (structs/dataset.py:46-57,114-119, utils/nn_funcs.py:122-129,
    utils/loss_funcs.py:117-129); bag_of_word_loss against the reference's own expand + NLLLoss formulation"""
    import torch
    from dss_vae_amd.structs import Dataset, PlainExample
    from dss_vae_amd.utils.loss_funcs import bag_of_word_loss
    from dss_vae_amd.utils.nn_funcs import padding_input
    f = tmp_path / "raw.txt"
    f.write_text("a b c\t( S x )\nd e\n")
    ds = Dataset.from_raw_file(str(f))
    assert [e.src for e in ds] == [["a", "b", "c"], ["d", "e"]] and ds.examples[0].tgt == ["(", "S", "x", ")"]
    assert ds.examples[1].tgt is None
    with pytest.raises(NotImplementedError):
        Dataset.from_raw_file(str(f), e_type="PTB")
    x, y = PlainExample.parse_for_control("a b\tc d e\n")
    assert x.src == ["a", "b"] and y.src == ["c", "d", "e"] and x.tgt is None
    assert padding_input([["a"], ["b", "c"]], "<pad>") == [["a", "<pad>"], ["b", "c"]]
    assert padding_input([["a"]], 0, max_len=3) == [["a", 0, 0]]
    g = torch.Generator().manual_seed(0)
    logp = torch.log_softmax(torch.randn(3, 1, 11, generator=g), -1)
    tgt = torch.tensor([[1, 4, 2, 0], [5, 0, 0, 0], [7, 8, 9, 10]])
    for red in ("sum", "mean"):
        crit = torch.nn.NLLLoss(ignore_index=0, reduction=red)
        want = crit(logp.squeeze(1).unsqueeze(1).expand(3, 4, 11).contiguous().view(-1, 11), tgt.view(-1))
        assert torch.allclose(bag_of_word_loss(logp, tgt, crit), want, rtol=1e-6, atol=1e-6)


def test_lr_helpers_and_yaml_args(tmp_path):
    """utils/model_utils.py:33-49 (with the `args` the reference forgets to pass), utils/config_funcs.py:26-33"""
    import math
    from argparse import Namespace
    from dss_vae_amd.utils.config_funcs import args_to_yaml, yaml_to_dict
    from dss_vae_amd.utils.model_utils import get_eval_info, get_lr_anneal, lr_transformer, update_lr
    a = Namespace(lr=0.001, anneal_steps=100, lr_schedule="anneal", d_model=512, warmup=4000)
    assert abs(get_lr_anneal(a, 50) - ((0.001 - 1e-5) * 0.5 + 1e-5)) < 1e-12 and get_lr_anneal(a, 500) == 1e-5
    assert abs(lr_transformer(a, 10) - 0.1 * 10 / math.sqrt(512) * 10 / (4000 * math.sqrt(4000))) < 1e-15

    class Opt:
        param_groups = [{"lr": 0.0}]
    assert update_lr(a, Opt, 49).param_groups[0]["lr"] == get_lr_anneal(a, 50)
    a.lr_schedule = "fixed"
    assert update_lr(a, Opt, 3).param_groups[0]["lr"] == 0.001
    a.lr_schedule = "transformer"
    assert update_lr(a, Opt, 9).param_groups[0]["lr"] == lr_transformer(a, 10)
    args_to_yaml(str(tmp_path / "a.yaml"), a)
    assert yaml_to_dict(str(tmp_path / "a.yaml"))["anneal_steps"] == 100
    d = get_eval_info(Namespace(exp_name="x", model_dir=str(tmp_path / "m")), Namespace(model_select="SyntaxVAE2"), "Test")
    assert d.endswith(os.path.join("SyntaxVAE2.x", "Test")) and os.path.isdir(d)


def test_metrics_tools(tmp_path):
    """metrics/tools.py:35-164 helpers: nested decode-result shapes, pair splitting, BLEU over results, corpus stats"""
    from dss_vae_amd.metrics import tools
    v = VocabEntry()
    for w in ("a", "b", "c"):
        v.add(w)
    exs = [PlainExample(src=["a", "b"], tgt=["c", "zzz"]), PlainExample(src=["c"], tgt=["a"])]
    sem, syn = tools.split_examples(exs)
    assert [e.src for e in sem] == [["a", "b"], ["c"]] and [e.src for e in syn] == [["c", "zzz"], ["a"]]
    assert all(e.tgt is None for e in sem + syn)
    assert tools.recovery_ref(syn, v, keep_origin=True) == [[["c", "zzz"]], [["a"]]]
    assert tools.recovery_ref(syn, v) == [[["c", "<unk>"]], [["a"]]]
    res = [[["a", "b", "c", "a", "b"]], [5]]
    assert tools.to_plain([res]) == [["a", "b", "c", "a", "b"]]
    assert tools.con_to_plain([[["x", "y"]]]) == ["x y"] and tools.reference_to_plain([[[[["x", "y"]]]]]) == ["x y"]
    assert abs(tools.get_bleu_score([[["a", "b", "c", "a", "b"]]], [res]) - 100.0) < 1e-9
    assert tools.get_bleu_score([[["c", "c", "c", "c", "c"]]], [res]) < 50.0
    assert [len(b) for b in tools.prepare_input(exs * 3, batch_size=4)] == [4, 2]
    assert tools.avg_length([["a"], ["a", "b", "c"]]) == 2.0
    same = tools.unigram_divergence([["a", "b"]], [["b", "a"]], v)
    diff = tools.unigram_divergence([["a", "a"]], [["c", "c"]], v)
    assert abs(same) < 1e-12 and diff > 0.05 and tools.unigram_divergence([["a"] * 10], [["c"] * 10], v, dtype="kl") > 0.5
    tj = tools.TagJudge(v)
    assert tj.is_tag("a") and not tj.is_tag("never-seen")
    f = str(tmp_path / "s.score")
    tools.write_append_result(["x"], f)
    tools.write_append_result([["y", "z"]], f)
    tools.write_append_result({"k": 1.5}, f)
    assert open(f).read().splitlines() == ["x", "y z", "k:1.500000"]


def test_rows_to_words_is_robust_id2word_row_by_row():
    """the vectorised token-matrix -> words conversion of the decode paths against the reference's per-token loop
    (utils/nn_funcs.py robust_id2word: cut at the first </s>, <s> and <pad> dropped): random matrices with the
    special ids anywhere, rows without </s>, </s> in column 0, an empty matrix"""
    import numpy as np
    from dss_vae_amd.structs.vocab import VocabEntry
    from dss_vae_amd.utils.nn_funcs import robust_id2word, rows_to_sentences, rows_to_words
    v = VocabEntry()
    for i in range(40):
        v.add("w%d" % i)
    rng = np.random.default_rng(0)
    for L in (1, 7, 35):
        rows = rng.integers(0, len(v), size=(23, L))
        rows[0, :] = 5                      # no </s> at all
        rows[1, 0] = v.eos_id               # empty sentence
        rows[2, :] = v.pad_id
        want = [robust_id2word(r, v) for r in rows.tolist()]
        assert rows_to_words(rows, v) == want
        assert rows_to_words(rows.tolist(), v) == want
        assert rows_to_sentences(rows, v) == [[[w], [len(w)]] for w in want]
    assert rows_to_words(np.zeros((0, 5), dtype=np.int64), v) == []
    v.add("late")                           # the table follows a vocabulary that grew
    assert rows_to_words(np.array([[len(v) - 1, v.eos_id]]), v) == [["late"]]
