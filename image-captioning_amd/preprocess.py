"""Vocabulary / caption encoding helpers (dense_img_cap_separate_models/preprocess.py:8-114).
The reference tokenises with nltk.word_tokenize (not installable here): treebank.py restates its Penn Treebank word rules
(contractions, punctuation, quotes, brackets: "man's" -> man + 's, "isn't" -> is + n't, held to NLTK's documented examples) and
APPROXIMATES the Punkt sentence splitter in front of them (trained parameters unavailable: a fixed abbreviation list).  Single-
phrase region captions tokenise identically; a caption with an interior period-final token that Punkt knows as an abbreviation
and treebank.py does not (or vice versa) gets an extra / a missing '.' token there, i.e. different ids for that caption."""
import numpy as np

from .treebank import word_tokenize  # noqa: F401  (re-exported: the reference imports it into this module's namespace)


def load_embeddings(file_name):
    """GloVe text file -> {word: float vector} (dense_img_cap/preprocess.py:30-40); keys lower-cased."""
    table = {}
    with open(file_name, 'r', encoding='utf-8') as doc:
        for line in doc:
            parts = line.rstrip('\n').lower().split(' ')
            if len(parts) > 1:
                table[parts[0]] = np.array(parts[1:], dtype=np.float64)
    return table


def tokenize_corpus(data_file, train, embeddings, min_count=15):
    """Vocabulary of the training regions (dense_img_cap/preprocess.py:43-56): tokens seen at least 15 times that have an
    embedding and are not punctuation."""
    import json
    from collections import Counter
    from string import punctuation
    train = set(train)
    counts = Counter()
    with open(data_file, 'r', encoding='utf-8') as doc:
        for image in json.load(doc):
            if image['id'] in train:
                for region in image['regions']:
                    counts.update(word_tokenize(region['phrase'].lower()))
    return {w for w, n in counts.items() if n >= min_count and w in embeddings and w not in punctuation}


def load_corpus(tokens, embeddings, embeddings_dim):
    """ids: 0 <unk>/pad (zero row), 1 <start>, 2 <end> (uniform(-0.5,0.5) rows), then the tokens."""
    id_to_word = {0: '<unk>', 1: '<start>', 2: '<end>'}
    word_to_id = {'<unk>': 0, '<start>': 1, '<end>': 2}
    matrix = np.zeros((len(tokens) + 3, embeddings_dim))
    matrix[1, :] = np.random.rand(embeddings_dim) - 0.5
    matrix[2, :] = np.random.rand(embeddings_dim) - 0.5
    for i, tok in enumerate(tokens):
        id_to_word[i + 3] = tok
        word_to_id[tok] = i + 3
        matrix[i + 3, :] = embeddings[tok]
    return word_to_id, id_to_word, matrix


def encode_word(word, word_to_id):
    return word_to_id.get(word, 0)


def encode_caption(caption, word_to_id):
    """v1: word ids, OOV dropped."""
    ids = [encode_word(t, word_to_id) for t in word_tokenize(caption.lower())]
    return np.array([i for i in ids if i != 0])


def encode_word_v2(word, word_to_id):
    vec = np.zeros(len(word_to_id))
    vec[word_to_id[word] if word in word_to_id else word_to_id.get('<UNK>', word_to_id.get('<unk>', 0))] = 1
    return vec


def encode_caption_v2(caption, word_to_id):
    """v2: one-hot rows, OOV (index 0) dropped... the reference drops rows whose element 0 is set."""
    rows = [encode_word_v2(t, word_to_id) for t in word_tokenize(caption.lower())]
    return np.array([r for r in rows if len(r) != 0 and r[0] != 1])


def decode_word(vec, id_to_word):
    return id_to_word[int(np.argmax(vec))]


def decode_caption(vector, id_to_word):
    return ''.join(decode_word(v, id_to_word) + ' ' for v in vector)
