"""Generates tests/golden/tokenizer_kat.json: the vocabulary, merge list and special tokens the reference's phoneme
tokenizer is configured with (data read from the reference's data/tokenizer.json) together with known answers from
the library the reference tokenizes with
(transformers.PreTrainedTokenizerFast, reference config.py:1184; call sites data.py:640-643).  TEST INFRASTRUCTURE;
run in the build container only:  python oracle/gen_tokenizer_kat.py"""
import json
import random
from pathlib import Path

from transformers import PreTrainedTokenizerFast

ROOT = Path(__file__).resolve().parent.parent
SRC = Path("/root/reference/data/tokenizer.json")
GOLD = ROOT / "tests" / "golden"


def main():
    ref = PreTrainedTokenizerFast(tokenizer_file=str(SRC))
    spec = json.loads(SRC.read_text(encoding="utf-8"))
    syms = [k for k in spec["model"]["vocab"] if not k.startswith("<")]
    random.seed(11)
    cases = ["wˈʌns ɐɡˈɛn, ˈɛksələnt wˈɜːk.", "", " ", "ðə kwˈɪk bɹˈaʊn fˈɑːks dʒˈʌmps ˌoʊvɚ ðə lˈeɪzi dˈɑːɡ.", "aaa", "ˈˈɛɛ",
             "hello 你好 <mask> x<bos>y</eos>", "ZQ@#", "ɸɹᵻv", "äɪnhɜː"]
    for _ in range(190):
        cases.append("".join(random.choice(syms) for _ in range(random.randint(1, 40))))
    post = spec["post_processor"]
    doc = {"vocab": spec["model"]["vocab"], "merges": spec["model"]["merges"], "unk_token": spec["model"]["unk_token"],
           "added_tokens": [{"id": t["id"], "content": t["content"]} for t in spec["added_tokens"]],
           "template_single": post["single"], "template_special_tokens": post["special_tokens"],
           "generator": "transformers.PreTrainedTokenizerFast(tokenizer_file=reference data/tokenizer.json).encode, "
                        "as reference config.py:1184 / data.py:640-643",
           "cases": [[c, ref.encode(c)] for c in cases]}
    (GOLD / "tokenizer_kat.json").write_text(json.dumps(doc, ensure_ascii=False), encoding="utf-8")
    print(len(cases), "cases")


if __name__ == "__main__":
    main()
