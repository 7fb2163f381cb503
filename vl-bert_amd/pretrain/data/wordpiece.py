"""BERT's WordPiece tokenizer for the data readers -- the surface the reference's datasets use (`vocab`, `basic_tokenizer.tokenize`,
`wordpiece_tokenizer.tokenize`, `tokenize`, `convert_tokens_to_ids` / `convert_ids_to_tokens`).

The reference takes it from a vendored third-party package (external/pytorch_pretrained_bert/tokenization.py, pytorch-pretrained-bert
0.6.x; not part of the hot path); the `transformers` build of this image ships only the Rust-backed tokenizer, which has no
word-level / piece-level split to hang whole-word masking on.  This is a restatement of the published algorithm (Devlin et al.,
google-research/bert tokenization): clean -> space out CJK -> whitespace split -> lower-case + strip combining marks -> split off every
punctuation character; then greedy longest-prefix matching against the vocabulary with '##' continuation pieces, words over 100
characters or with an unmatchable remainder -> [UNK].  Pinned to the reference's tokenizer on the fixture corpus and a set of awkward
strings (tests/test_data_cpu.py, golden made by oracle/make_data_golden.py).
"""
import collections
import unicodedata

NEVER_SPLIT = ("[UNK]", "[SEP]", "[PAD]", "[CLS]", "[MASK]")
_CJK = ((0x4E00, 0x9FFF), (0x3400, 0x4DBF), (0x20000, 0x2A6DF), (0x2A700, 0x2B73F), (0x2B740, 0x2B81F), (0x2B820, 0x2CEAF),
        (0xF900, 0xFAFF), (0x2F800, 0x2FA1F))


def _space(ch):
    return ch in " \t\n\r" or unicodedata.category(ch) == "Zs"


def _control(ch):
    return ch not in "\t\n\r" and unicodedata.category(ch).startswith("C")


def _punct(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:      # every non-alphanumeric ASCII symbol counts
        return True
    return unicodedata.category(ch).startswith("P")


class BasicTokenizer:
    def __init__(self, do_lower_case=True, never_split=NEVER_SPLIT):
        self.do_lower_case, self.never_split = do_lower_case, never_split

    def tokenize(self, text):
        chars = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _control(ch):
                continue
            if _space(ch):
                chars.append(" ")
            elif any(lo <= cp <= hi for lo, hi in _CJK):
                chars.append(" " + ch + " ")
            else:
                chars.append(ch)
        words = []
        for tok in "".join(chars).split():
            if tok in self.never_split:
                words.append(tok)
                continue
            if self.do_lower_case:
                tok = "".join(c for c in unicodedata.normalize("NFD", tok.lower()) if unicodedata.category(c) != "Mn")
            run = ""
            for c in tok:
                if _punct(c):
                    if run:
                        words.append(run)
                    words.append(c)
                    run = ""
                else:
                    run += c
            if run:
                words.append(run)
        return " ".join(words).split()


class WordpieceTokenizer:
    def __init__(self, vocab, unk_token="[UNK]", max_input_chars_per_word=100):
        self.vocab, self.unk_token, self.max_input_chars_per_word = vocab, unk_token, max_input_chars_per_word

    def tokenize(self, text):
        out = []
        for word in text.split():
            if len(word) > self.max_input_chars_per_word:
                out.append(self.unk_token)
                continue
            pieces, start = [], 0
            while start < len(word):
                end = len(word)
                while end > start:
                    cand = ("##" if start else "") + word[start:end]
                    if cand in self.vocab:
                        break
                    end -= 1
                if end == start:
                    pieces = None
                    break
                pieces.append(cand)
                start = end
            out.extend(pieces if pieces is not None else [self.unk_token])
        return out


class BertTokenizer:
    def __init__(self, vocab_file, do_lower_case=True, max_len=None, never_split=NEVER_SPLIT):
        self.vocab = collections.OrderedDict()
        with open(vocab_file, "r", encoding="utf-8") as f:
            for index, line in enumerate(f):      # one token per line, id = line number
                self.vocab[line.strip()] = index
        self.ids_to_tokens = collections.OrderedDict((i, t) for t, i in self.vocab.items())
        self.basic_tokenizer = BasicTokenizer(do_lower_case, never_split)
        self.wordpiece_tokenizer = WordpieceTokenizer(self.vocab)
        self.max_len = max_len if max_len is not None else int(1e12)

    def tokenize(self, text):
        return [p for w in self.basic_tokenizer.tokenize(text) for p in self.wordpiece_tokenizer.tokenize(w)]

    def convert_tokens_to_ids(self, tokens):
        return [self.vocab[t] for t in tokens]

    def convert_ids_to_tokens(self, ids):
        return [self.ids_to_tokens[i] for i in ids]
