"""Corpus BLEU with the semantics of nltk.translate.bleu_score.corpus_bleu at its defaults (uniform
4-gram weights, no smoothing, brevity penalty against the closest reference length), which the
reference calls through BleuScoreMetric (dss_vae/metrics/bleu_scorer.py:10-66).  nltk is not available
on the GPU box, so the arithmetic is restated here: modified n-gram precisions are pooled over the
corpus (Papineni et al. 2002); any zero numerator gives BLEU 0."""
import math
from collections import Counter


def _ngrams(tokens, n):
    return Counter(tuple(tokens[i:i + n]) for i in range(len(tokens) - n + 1))


def corpus_bleu(list_of_references, hypotheses, max_n=4):
    """list_of_references: list (per segment) of lists of token lists; hypotheses: list of token lists."""
    assert len(list_of_references) == len(hypotheses)
    num, den = [0] * max_n, [0] * max_n
    hyp_len = ref_len = 0
    for refs, hyp in zip(list_of_references, hypotheses):
        hyp_len += len(hyp)
        ref_len += min((abs(len(r) - len(hyp)), len(r)) for r in refs)[1]
        for n in range(1, max_n + 1):
            counts = _ngrams(hyp, n)
            best = Counter()
            for r in refs:
                for g, c in _ngrams(r, n).items():
                    if c > best[g]:
                        best[g] = c
            num[n - 1] += sum(min(c, best[g]) for g, c in counts.items())
            den[n - 1] += max(1, sum(counts.values()))
    if hyp_len == 0 or num[0] == 0:
        return 0.0
    if any(x == 0 for x in num):
        return 0.0
    bp = 1.0 if hyp_len > ref_len else math.exp(1.0 - float(ref_len) / hyp_len)
    return bp * math.exp(sum(math.log(float(a) / b) for a, b in zip(num, den)) / max_n)


def evaluate_lines(pred_lines, gold_lines_list):
    """evaluate_file on lines that were never written: same tokenisation (strip, split on one blank), so a
    rank that keeps its dev decode in memory scores exactly what the writing rank reads back."""
    hyps = [line.strip().split(" ") for line in pred_lines]
    refs = [[] for _ in hyps]
    for lines in gold_lines_list:
        for i, line in enumerate(lines):
            refs[i].append(line.strip().split(" "))
    return 100.0 * corpus_bleu(refs, hyps)


def evaluate_file(pred_file, gold_files):
    """BleuScoreMetric.evaluate_file: one hypothesis per line, references line-aligned; x100."""
    with open(pred_file) as f:
        hyps = [line.strip().split(" ") for line in f.readlines()]
    refs = [[] for _ in hyps]
    for path in (gold_files if isinstance(gold_files, list) else [gold_files]):
        with open(path) as f:
            for i, line in enumerate(f.readlines()):
                refs[i].append(line.strip().split(" "))
    return 100.0 * corpus_bleu(refs, hyps)
