"""Entry point of the joint model, mirroring dense_img_cap/train_dense_captions.py (DenseCapConfig :18-41,
VisualGenomeDataset :44-115, __main__ :118-210): build the vocabulary, the two datasets and the model, load the
COCO-pretrained backbone and the separately trained caption head by name, train layers="no_backbone".

Multi-GPU: launch one process per GPU (python -m torch.distributed.run --nproc-per-node N train_dense_captions.py) and
set DenseCapConfig.GPU_COUNT = N; the model is then wrapped in ParallelModel like the reference's build() does."""
import json
import os
import pickle
import time

import numpy as np

from .config import Config
from .dense_model import DenseImageCapRCNN
from .preprocess import encode_caption, load_corpus, load_embeddings, tokenize_corpus
from .text_generation_model_v2 import pad_sequences
from .utils import Dataset


class DenseCapConfig(Config):
    NAME = "dense image captioning"
    GPU_COUNT = 1
    IMAGES_PER_GPU = 1
    STEPS_PER_EPOCH = 1000
    VALIDATION_STEPS = 50
    PADDING_SIZE = 15

    def __init__(self, vocab_size, embedding_weights):
        super(DenseCapConfig, self).__init__()
        self.VOCABULARY_SIZE = vocab_size
        self.EMBEDDING_WEIGHTS = embedding_weights
        self.EMBEDDING_SIZE = embedding_weights.shape[1]


class VisualGenomeDataset(Dataset):
    def __init__(self, words_to_ids, padding_size):
        super(VisualGenomeDataset, self).__init__()
        self.word_to_id = words_to_ids
        self.padding_size = padding_size

    def load_visual_genome(self, data_dir, image_ids, image_meta_file, data_file):
        with open(data_file, 'r', encoding='utf-8') as doc:
            regions = {x['id']: x['regions'] for x in json.load(doc)}
        with open(image_meta_file, 'r', encoding='utf-8') as doc:
            meta = {x['image_id']: x for x in json.load(doc)}
        for i in image_ids:
            self.add_image("VisualGenome", image_id=i, path=os.path.join(data_dir, '{}.jpg'.format(i)),
                           width=meta[i]['width'], height=meta[i]['height'],
                           rois=[[d['y'], d['x'], d['y'] + d['height'], d['x'] + d['width']] for d in regions[i]],
                           captions=[[d['phrase']] for d in regions[i]])

    def image_reference(self, image_id):
        return "https://cs.stanford.edu/people/rak248/VG_100K/{}.jpg".format(self.image_info[image_id]["id"])

    def load_captions_and_rois(self, image_id):
        """rois [N,4]; captions float32 [N,T] = [1 <start>, ids..., 2 <end>, 0 pad], truncated to T (:88-103)."""
        info = self.image_info[image_id]
        T = self.padding_size
        rois, caps = [], []
        for roi, caption in zip(info['rois'], info['captions']):
            cap = self.encode_region_caption(caption[0])
            if cap.size != 0:
                rois.append(roi)
                caps.append(np.hstack((np.array(1), cap[:T - 2], np.array(2))))
        captions = pad_sequences(caps, maxlen=T, padding='post', dtype='float').astype(np.float32)
        return np.array(rois), captions

    def load_original_captions_and_rois(self, image_id):
        info = self.image_info[image_id]
        return np.array(info['rois']), info['captions']

    def encode_region_caption(self, caption):
        return encode_caption(caption, self.word_to_id)


def load_vocabulary(cache_dir, embeddings_file, data_file, train_image_ids):
    """id_to_word / word_to_id / embedding_matrix, cached as pickles like the reference (:143-166)."""
    files = [os.path.join(cache_dir, n + '.pickle') for n in ('id_to_word', 'word_to_id', 'embedding_matrix')]
    if all(os.path.exists(f) for f in files):
        return [pickle.load(open(f, 'rb')) for f in files]
    embeddings = load_embeddings(embeddings_file)
    tokens = tokenize_corpus(data_file, train_image_ids, embeddings)
    word_to_id, id_to_word, matrix = load_corpus(sorted(tokens), embeddings, 300)
    pad = (-matrix.shape[0]) % 4                        # the GPU kernels want 16-byte rows: pad the vocabulary with unused ids
    matrix = np.concatenate([matrix, np.zeros((pad, matrix.shape[1]))])
    for i in range(pad):
        id_to_word[len(id_to_word)] = '<pad%d>' % i
    os.makedirs(cache_dir, exist_ok=True)
    for f, obj in zip(files, (id_to_word, word_to_id, matrix)):
        with open(f, 'wb') as fh:
            pickle.dump(obj, fh, protocol=pickle.HIGHEST_PROTOCOL)
    return id_to_word, word_to_id, matrix


def main(root_dir=None, init_with='coco', epochs=100):
    from .parallel_model import ParallelModel, init_process_group_from_env
    rank, world, _ = init_process_group_from_env()
    root_dir = root_dir or os.getcwd()
    model_dir = os.path.join(root_dir, "logs_dense_img_cap")
    coco_model_path = os.path.join(root_dir, "../mask_rcnn_coco.npz")
    image_meta_file_path = '../dataset/image_data.json'
    data_file_path = '../dataset/region_descriptions.json'
    with open(image_meta_file_path, 'r', encoding='utf-8') as f:
        image_ids_list = [m['image_id'] for m in json.load(f)]
    train_image_ids, val_image_ids = image_ids_list[:90000], image_ids_list[90000:100000]
    id_to_word, word_to_id, embedding_matrix = load_vocabulary('../dataset/dense_img_cap', '../dataset/glove.6B.300d.txt',
                                                               data_file_path, train_image_ids)
    config = DenseCapConfig(len(id_to_word), embedding_matrix)
    config.GPU_COUNT = world
    if rank == 0:
        config.display()
    datasets = []
    for ids in (train_image_ids, val_image_ids):
        ds = VisualGenomeDataset(word_to_id, config.PADDING_SIZE)
        ds.load_visual_genome('../dataset/visual genome/', ids, image_meta_file_path, data_file_path)
        ds.prepare()
        datasets.append(ds)
    model = DenseImageCapRCNN(mode="training", config=config, model_dir=model_dir)
    if init_with == "last":
        model.load_weights(model.find_last()[1], by_name=True)
    else:
        model.load_weights(coco_model_path, by_name=True)
        model.load_weights('../dense_img_cap_separate_models/models/model-47-1.74.npz', by_name=True)
    if world > 1:
        model = ParallelModel(model, world)
    if rank == 0:
        print(model.summary())
    start_time = time.time()
    model.train(datasets[0], datasets[1], learning_rate=config.LEARNING_RATE, epochs=epochs, layers="no_backbone")
    print(time.time() - start_time)


if __name__ == '__main__':
    main()
