"""XLM-R style token ids over a sentencepiece model -- what the reference gets from
`transformers.XLMRobertaTokenizer("pretrain_weights/beit3.spm")` (`simvg/datasets/pipelines/loading.py:74-77,157-182`; the
sentencepiece-backed tokenizer of transformers 4.x, which the transformers 5.x of this image no longer ships).

The algorithm is the published fairseq alignment of that tokenizer:
    id 0 <s>, 1 <pad>, 2 </s>, 3 <unk>; sentencepiece piece p (p > 0) -> p + 1; sentencepiece's own <unk> (0) -> 3;
    vocab_size = len(spm) + 1 (offset) + 1 (<mask>, the last id).
Parity UNPINNED against transformers 4.x itself (not installable here) and against BEiT-3's `beit3.spm` (not in this image);
`tests/test_datasets_cpu.py` checks the alignment rules on a sentencepiece model trained in the test."""
import os


class XLMRTokenizer:
    bos_token, pad_token, eos_token, unk_token, mask_token = "<s>", "<pad>", "</s>", "<unk>", "<mask>"
    bos_token_id, pad_token_id, eos_token_id, unk_token_id = 0, 1, 2, 3
    fairseq_offset = 1

    def __init__(self, vocab_file):
        import sentencepiece as spm
        if not os.path.isfile(vocab_file):
            raise FileNotFoundError(f"sentencepiece model {vocab_file!r} not found (BEiT-3 ships it as beit3.spm; pass "
                                    "spm_path=... to LoadImageAnnotationsFromFile)")
        self.sp_model = spm.SentencePieceProcessor()
        self.sp_model.Load(vocab_file)
        self._special = {self.bos_token: 0, self.pad_token: 1, self.eos_token: 2, self.unk_token: 3}
        self.mask_token_id = len(self.sp_model) + self.fairseq_offset
        self._special[self.mask_token] = self.mask_token_id

    @property
    def vocab_size(self):
        return len(self.sp_model) + self.fairseq_offset + 1

    def tokenize(self, text):
        return self.sp_model.encode(text, out_type=str)

    def convert_tokens_to_ids(self, tokens):
        if isinstance(tokens, str):
            return self._id(tokens)
        return [self._id(t) for t in tokens]

    def _id(self, token):
        if token in self._special:
            return self._special[token]
        p = self.sp_model.PieceToId(token)
        return p + self.fairseq_offset if p else self.unk_token_id

    def encode_expression(self, text, max_token):
        """<s> pieces </s> padded to max_token -> (ids, padding_mask) with mask 1 = pad (loading.py:157-182)"""
        ids = self.convert_tokens_to_ids(self.tokenize(text))
        if not ids:
            raise RuntimeError("The text segment should contains at least one tokens!")
        ids = [self.bos_token_id] + ids[:max_token - 2] + [self.eos_token_id]
        n = len(ids)
        return ids + [self.pad_token_id] * (max_token - n), [0] * n + [1] * (max_token - n)
