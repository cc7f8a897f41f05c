"""Dictionary pre-training with the names and the on-disk format of the reference's
``gccNMF/realtime/gccNMFPretraining.py`` (:43-87): dictionaries are float32 ``(F, K)`` arrays cached as
``<DATA_DIR>/pretrainedW/W_<K>.npy``, so the reference's real-time application can consume dictionaries trained here.

The training itself is ``performKLNMF`` on the GPU (gcc_nmf_amd.gccNMFFunctions, csrc/nmf.hip).  The reference trains on
``data/chimeTrainSet.npy``, which is not part of its checkout; when that file is absent the training matrix is built from
the magnitude spectrograms of the ``*_mix.wav`` files found in DATA_DIR (``buildTrainingSet``).
"""
import glob
import logging
import os
from collections import OrderedDict
from os.path import exists, join

import numpy as np

from .gccNMFFunctions import performKLNMF, computeComplexMixtureSpectrogram, hanning
from .wavfile import wavread

DATA_DIR = os.environ.get('GCCNMF_DATA_DIR', join(os.getcwd(), 'data'))        # gccNMF/defs.py:37
PRETRAINED_W_DIR = join(DATA_DIR, 'pretrainedW')
PRETRAINED_W_PATH_TEMPLATE = join(PRETRAINED_W_DIR, 'W_%d.npy')
SPARSITY_ALPHA = 0
NUM_PRELEARNING_ITERATIONS = 100
CHIME_DATASET_PATH = join(DATA_DIR, 'chimeTrainSet.npy')


def configure(dataDir):
    """Point the module at another data directory (the reference reads GCCNMF_DATA_DIR once at import)."""
    global DATA_DIR, PRETRAINED_W_DIR, PRETRAINED_W_PATH_TEMPLATE, CHIME_DATASET_PATH
    DATA_DIR = dataDir
    PRETRAINED_W_DIR = join(DATA_DIR, 'pretrainedW')
    PRETRAINED_W_PATH_TEMPLATE = join(PRETRAINED_W_DIR, 'W_%d.npy')
    CHIME_DATASET_PATH = join(DATA_DIR, 'chimeTrainSet.npy')


def buildTrainingSet(windowSize=1024, hopSize=512, wavPaths=None):
    """(F, N) float32 training matrix: |STFT| of both channels of every mixture, frames concatenated in file order."""
    wavPaths = sorted(glob.glob(join(DATA_DIR, '*_mix.wav'))) if wavPaths is None else list(wavPaths)
    if not wavPaths:
        raise IOError('no training data: neither %s nor any *_mix.wav in %s' % (CHIME_DATASET_PATH, DATA_DIR))
    blocks = []
    for path in wavPaths:
        stereoSamples, _ = wavread(path)
        X = computeComplexMixtureSpectrogram(stereoSamples, windowSize, hopSize, hanning)
        blocks.append(np.concatenate(np.abs(X), axis=-1))
    return np.concatenate(blocks, axis=-1).astype(np.float32)


def getOrderedDictionary(W):
    """gccNMFPretraining.py:60-66: atoms sorted by spectral centroid (ascending)."""
    numFreq, _ = W.shape
    spectralCentroids = np.sum(np.arange(numFreq)[:, np.newaxis] * W, axis=0, keepdims=True) / np.sum(W, axis=0, keepdims=True)
    orderedAtomIndexes = np.argsort(np.squeeze(spectralCentroids))
    return np.squeeze(W[:, orderedAtomIndexes])


def loadPretrainedW(dictionarySize, retrainW=False, windowSize=1024, hopSize=512):
    """gccNMFPretraining.py:68-87: load ``W_<K>.npy`` or train it (100 KL-NMF iterations, alpha 0, seed 0) and cache it."""
    pretrainedWFilePath = PRETRAINED_W_PATH_TEMPLATE % dictionarySize
    logging.info('GCCNMFPretraining: Loading pretrained W (size %d): %s' % (dictionarySize, pretrainedWFilePath))
    if exists(pretrainedWFilePath) and not retrainW:
        return np.load(pretrainedWFilePath)
    if retrainW:
        logging.info('GCCNMFPretraining: Retraining W, saving as %s...' % pretrainedWFilePath)
    else:
        logging.info('GCCNMFPretraining: Pretrained W not found at %s, creating...' % pretrainedWFilePath)
    trainV = np.load(CHIME_DATASET_PATH) if exists(CHIME_DATASET_PATH) else buildTrainingSet(windowSize, hopSize)
    W, _ = performKLNMF(trainV, dictionarySize, numIterations=NUM_PRELEARNING_ITERATIONS, sparsityAlpha=SPARSITY_ALPHA, epsilon=1e-16,
                        seedValue=0)
    os.makedirs(PRETRAINED_W_DIR, exist_ok=True)
    np.save(pretrainedWFilePath, W)
    return W


def getDictionariesW(windowSize, dictionarySizes, ordered=False):
    """gccNMFPretraining.py:43-58."""
    fftSize = windowSize // 2 + 1
    dictionariesW = OrderedDict([
        ('Pretrained', OrderedDict([(k, loadPretrainedW(k, windowSize=windowSize, hopSize=windowSize // 2)) for k in dictionarySizes])),
        ('Random', OrderedDict([(k, np.random.rand(fftSize, k).astype('float32')) for k in dictionarySizes]))])
    if not ordered:
        return dictionariesW
    orderedDictionariesW = OrderedDict()
    for dictionaryType, dictionaries in dictionariesW.items():
        orderedDictionariesW[dictionaryType] = OrderedDict((k, getOrderedDictionary(W)) for k, W in dictionaries.items())
    return orderedDictionariesW
