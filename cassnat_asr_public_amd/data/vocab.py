"""Token table with the reference's conventions (src/data/vocab.py:4-43): ids 0..3 are blank/sos/eos/unk, every
further distinct token of the vocabulary file gets the next id in order of first appearance.  A line is either a
single token or "<key> tok tok ..." (the first field is dropped when a line has more than one)."""


class Vocab(object):
    SPECIALS = ("blank", "sos", "eos", "unk")

    def __init__(self, vocab_file, rank=0):
        self.vocab_file, self.rank = vocab_file, rank
        self.word2index = {w: i for i, w in enumerate(self.SPECIALS)}
        self.index2word = {i: w for i, w in enumerate(self.SPECIALS)}
        self.word2count = {}
        self.n_words = len(self.SPECIALS)
        self.read_lang()

    def add_word(self, word):
        if word in self.word2index:
            self.word2count[word] = self.word2count.get(word, 0) + 1
            return
        self.word2index[word] = self.n_words
        self.index2word[self.n_words] = word
        self.word2count[word] = 1
        self.n_words += 1

    def add_sentence(self, sentence):
        for word in sentence.split(" "):
            self.add_word(word)

    def read_lang(self):
        if self.rank == 0:
            print("Reading vocabulary from {}".format(self.vocab_file))
        with open(self.vocab_file, "r") as f:
            for raw in f:
                fields = raw.strip().split(" ")
                self.add_sentence(" ".join(fields[1:]) if len(fields) > 1 else fields[0])
        if self.rank == 0:
            print("Vocabulary size is {}".format(self.n_words))
