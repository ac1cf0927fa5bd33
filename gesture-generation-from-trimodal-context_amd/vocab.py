"""Minimal word / speaker vocabulary (the role of model/vocab.py:8-84 on the hot path).

PoseGenerator only needs `n_words` (rows of the speaker embedding) and the isinstance() gate at
multimodal_context_net.py:87; checkpoints pickle the object, so the attribute names follow the reference's.
fastText loading (vocab.py:69-84) is text preprocessing and out of scope.
"""


class Vocab:
    PAD_token, SOS_token, EOS_token, UNK_token = 0, 1, 2, 3

    def __init__(self, name, insert_default_tokens=True):
        self.name = name
        self.trimmed = False
        self.word_embedding_weights = None
        self.reset_dictionary(insert_default_tokens)

    def reset_dictionary(self, insert_default_tokens=True):
        self.word2index, self.word2count = {}, {}
        if insert_default_tokens:
            self.index2word = {0: "<PAD>", 1: "<SOS>", 2: "<EOS>", 3: "<UNK>"}
        else:
            self.index2word = {self.UNK_token: "<UNK>"}
        self.n_words = len(self.index2word)

    def index_word(self, word):
        if word in self.word2index:
            self.word2count[word] += 1
            return
        self.word2index[word] = self.n_words
        self.word2count[word] = 1
        self.index2word[self.n_words] = word
        self.n_words += 1

    def get_word_index(self, word):
        return self.word2index.get(word, self.UNK_token)

    @classmethod
    def speakers(cls, n_speakers):
        """Speaker vocabulary with ids 1..n_speakers-1 (id 0 unused), as lmdb_data_loader.py:178 builds it."""
        v = cls("vid", insert_default_tokens=False)
        for i in range(n_speakers - 1):
            v.index_word(f"spk{i:05d}")
        return v
