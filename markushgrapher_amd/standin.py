"""Stand-in tokenizer vocabulary for the main model's INPUT side (benchmarks and tests only).

The reference tokenises OCR words with the UDOP sentencepiece model (33k pieces), which is not available offline.  Stock
`UdopTokenizer(vocab=[(piece, score), ...])` accepts a Unigram vocabulary directly, so the stock tokenizer CLASS (pre-tokenizer, pair
template, box handling - everything `processor(text=, text_pair=, boxes=)` does, ref: utils/common.py:34-42) runs unchanged on a small
deterministic vocabulary.  Ids stay below 500 so that the tiny parity model (vocab 500) can embed them.
"""
_WORDS = ["Question", "Answering", "What", "markush", "structure", "is", "in", "the", "image", "alkyl", "group", "hydrogen", "atom", "halogen",
          "represents", "wherein", "and", "or", "same", "different", "each", "may", "be", "methyl", "ethyl", "phenyl", "alkoxy", "C1-C6", "R1", "R2",
          "R3", "R4", "OH", "NH", "Cl", "Br", "Me", "Et", "Ph", "Ar", "Het", "ring", "aryl", "from", "selected", "of", "a", "an"]
_CHARS = list("abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789.,;:=-()[]/?+*'")


def udop_standin_vocab():
    v = [("<pad>", 0.0), ("</s>", 0.0), ("<unk>", 0.0), ("▁", -2.0)]
    seen = {p for p, _ in v}

    def add(piece, score):
        if piece not in seen:
            seen.add(piece)
            v.append((piece, score - 0.01 * len(v)))
    for w in _WORDS:
        add("▁" + w, -4.0)
    for w in _WORDS:
        add(w, -6.0)
    for c in _CHARS:
        add(c, -9.0)
    for c in _CHARS:
        add("▁" + c, -9.5)
    assert len(v) < 500 and len({p for p, _ in v}) == len(v)
    return v


def make_udop_tokenizer():
    from transformers import UdopTokenizer
    return UdopTokenizer(vocab=udop_standin_vocab())


