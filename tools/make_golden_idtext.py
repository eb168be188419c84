"""Mints tests/golden/host_idtext.json: inputs and outputs of the REFERENCE's own `MarkushTokenizer.decode_plus_decode_other_tokens`
(/root/reference/markushgrapher/core/common/markush_tokenizer.py:615-670), executed unmodified in the build container.

The file is loaded by path; `rdkit` and `SmilesPE.pretokenizer`, which it imports at module level and which this image lacks, are
stubbed with EMPTY modules - the decode method touches neither.  The instance is made with `object.__new__` (the constructor reads
vocabulary JSON files of the training data set, which are not in the tree) and given exactly the attributes the method reads:
`tokenizer.convert_ids_to_tokens`, `vocabulary`, `vocabulary_inverse`, `encode_index`.  The id -> token table is a stand-in for the
UDOP sentencepiece vocabulary (not available offline) with every token class the method distinguishes: ordinary pieces with and
without the word-start marker, a bare marker, <loc_N>, <other_N> inside and outside the Markush vocabulary, `<i>` / `</i>`.
Only data (inputs / outputs) is written.
    python tools/make_golden_idtext.py
"""
import importlib.util
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/markushgrapher/core/common/markush_tokenizer.py"
SP = "▁"

MARKUSH = ["<cxsmi>", "</cxsmi>", "<r>", "</r>", "<markush>", "</markush>", "<stable>", "</stable>", "<n>", "<ns>", "<i>", "</i>",
           "C", "N", "O", "c", "(", ")", "=", "1", "[R1]", "Cl", "*"]
FIRST = 100


def token_table():
    toks = ["<pad>", "</s>", "<unk>", SP, SP + "alkyl", "group", SP + "R", "1", ":", SP + "H", ",", SP + "Me", "thyl", "<loc_12>",
            "<loc_499>", SP + "C", "1-6", "</s>x", "other", SP + "other", "loc", "<x>", "2"]
    toks += [f"<other_{i}>" for i in range(FIRST - 2, FIRST + len(MARKUSH) + 3)]     # two below and three above the Markush range
    return toks


class StandInTokenizer:
    def __init__(self, toks):
        self.toks = toks

    def convert_ids_to_tokens(self, ids):
        return [self.toks[int(i)] for i in ids]


def load_ref_class():
    for name in ("rdkit", "SmilesPE", "SmilesPE.pretokenizer"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["rdkit"].Chem = types.ModuleType("rdkit.Chem")
    sys.modules["SmilesPE.pretokenizer"].atomwise_tokenizer = lambda s: list(s)
    spec = importlib.util.spec_from_file_location("ref_markush_tokenizer", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.MarkushTokenizer


def make(cls, toks, encode_index):
    t = object.__new__(cls)
    t.tokenizer = StandInTokenizer(toks)
    t.vocabulary = {v: f"<other_{FIRST + i}>" for i, v in enumerate(MARKUSH)}
    t.vocabulary_inverse = {f"<other_{FIRST + i}>": v for i, v in enumerate(MARKUSH)}
    t.encode_index = encode_index
    return t


def cases(toks):
    ix = {t: i for i, t in enumerate(toks)}
    o = lambda s: ix[f"<other_{FIRST + MARKUSH.index(s)}>"]
    seqs = [
        [o("<markush>"), o("<cxsmi>"), o("C"), o("<i>"), ix["1"], o("</i>"), o("C"), o("<i>"), ix["2"], o("</i>"), o("("), o("="), o("O"), o(")"),
         o("[R1]"), o("</cxsmi>"), o("<stable>"), ix[SP + "R"], ix["1"], ix[":"], ix[SP + "H"], ix[","], ix[SP + "Me"], ix["thyl"], o("<n>"),
         ix[SP + "C"], ix["1-6"], ix[SP + "alkyl"], ix["group"], o("</stable>"), o("</markush>")],
        [ix[SP + "alkyl"], ix["group"], ix[SP + "alkyl"], ix["group"]],
        [ix[SP + "alkyl"], ix["<loc_12>"], ix["group"], ix["<loc_499>"]],
        [ix["group"], o("<cxsmi>")],
        [ix[f"<other_{FIRST - 1}>"], ix[f"<other_{FIRST + len(MARKUSH) + 1}>"], ix["group"]],
        [ix[SP], ix["group"], ix[SP], ix[SP + "H"]],
        [o("C"), o("<i>"), ix["1"], ix["2"], ix[SP + "H"], o("</i>"), o("C")],
        [o("<i>"), ix["1"]],                                     # unterminated index
        [o("</i>"), o("C"), o("</i>")],
        [ix["other"], ix[SP + "other"], ix["loc"], ix["<x>"], ix["2"]],
        [ix["<unk>"], ix["</s>x"], ix["</s>"], ix["<pad>"]],
        [],
        [o("Cl"), ix[SP + "H"], o("*"), ix["1"], o("1")],
    ]
    return seqs


def main():
    cls = load_ref_class()
    toks = token_table()
    out = {"tokens": toks, "first_other": FIRST, "markush_vocabulary": MARKUSH, "cases": []}
    for ei in (False, True):
        t = make(cls, toks, ei)
        for ids in cases(toks):
            out["cases"].append({"encode_index": ei, "ids": ids, "text": t.decode_plus_decode_other_tokens(ids)})
    path = os.path.join(ROOT, "tests", "golden", "host_idtext.json")
    with open(path, "w") as f:
        json.dump(out, f, ensure_ascii=False, indent=1)
    print("wrote", path, len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
