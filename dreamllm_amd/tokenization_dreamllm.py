"""Special-token constants of the reference (omni/models/dreamllm/tokenization_dreamllm.py:61-94).  Only the constants:
their ids drive the multimodal splice; the sentencepiece tokenizer itself is CPU text preprocessing (out of scope)."""
DEFAULT_BOS_TOKEN = "<s>"
DEFAULT_EOS_TOKEN = "</s>"
DEFAULT_UNK_TOKEN = "<unk>"
DEFAULT_PAD_TOKEN = "[PAD]"

DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_IMAGE_PATCH_TOKEN = "<im_patch>"
DEFAULT_IMAGE_START_TOKEN = "<im_start>"
DEFAULT_IMAGE_END_TOKEN = "<im_end>"

DEFAULT_DREAM_TOKEN = "<dream>"
DEFAULT_DREAM_PATCH_TOKEN = "<dream_patch>"
DEFAULT_DREAM_START_TOKEN = "<dream_start>"
DEFAULT_DREAM_END_TOKEN = "<dream_end>"

additional_special_tokens = [
    DEFAULT_IMAGE_TOKEN,
    DEFAULT_IMAGE_PATCH_TOKEN,
    DEFAULT_IMAGE_START_TOKEN,
    DEFAULT_IMAGE_END_TOKEN,
    DEFAULT_DREAM_TOKEN,
    DEFAULT_DREAM_START_TOKEN,
    DEFAULT_DREAM_END_TOKEN,
]

special_tokens_dict = dict(
    bos_token=DEFAULT_BOS_TOKEN,
    eos_token=DEFAULT_EOS_TOKEN,
    unk_token=DEFAULT_UNK_TOKEN,
    pad_token=DEFAULT_PAD_TOKEN,
    additional_special_tokens=additional_special_tokens,
)


def default_special_tokens2ids(base_vocab: int = 32000) -> dict:
    """ids as produced by LlamaTokenizer.add_special_tokens(special_tokens_dict) on a 32000-token LLaMA vocabulary
    (projects/dreamllm/train.py:74-96): [PAD] first, then the additional special tokens in list order."""
    ids = {DEFAULT_BOS_TOKEN: 1, DEFAULT_EOS_TOKEN: 2, DEFAULT_UNK_TOKEN: 0, DEFAULT_PAD_TOKEN: base_vocab}
    ids["additional_special_tokens"] = {t: base_vocab + 1 + i for i, t in enumerate(additional_special_tokens)}
    return ids
