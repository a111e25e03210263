#!/usr/bin/env python
"""Golden vectors for the PQ encoder.  The reference's TrainedPQEncoder.encode_multi
(extraction/descriptor_PQ.py:19-27) is a loop over sub-spaces around scipy.cluster.vq.vq; descriptor_PQ.py itself is Python 2
(its import of template_2 fails here), so the vectors are made by calling the SAME third-party routine the reference calls, on
the reference's codebook file, with the dtypes the reference's tool uses (float32 codewords, descriptor_PQ.py:323; float32
descriptors).  Run in the build container:  python tests/golden/make_golden_pq.py   (scipy version recorded in the file)."""
import os
import numpy as np
import scipy
from scipy.cluster.vq import vq

HERE = os.path.dirname(os.path.abspath(__file__))
raw = open(os.path.join(HERE, "codebook_EmbeddingSize_96_stride_16_subdim_6.dat"), "rb").read()
M, K, D = np.frombuffer(raw[:6], "<i2")
words = np.frombuffer(raw[6:], "<f4").reshape(M, K, D)

rng = np.random.default_rng(20240928)
n = 512
des = rng.standard_normal((n, M * D)).astype(np.float32)
des /= np.linalg.norm(des, axis=1, keepdims=True)                 # unit-norm embeddings, as the extraction network emits
des[:64] = np.concatenate([words[m][rng.integers(0, K, 64)] for m in range(M)], axis=1)   # exact codewords: distance 0
des[64:96] *= 1.73                                                 # the scale SURVEY §8d uses for minutiae descriptors
codes = np.empty((n, M), np.uint8)
for m in range(M):                                                 # encode_multi, descriptor_PQ.py:25-26
    codes[:, m], _ = vq(des[:, m * D:(m + 1) * D], words[m])
np.savez_compressed(os.path.join(HERE, "golden_pq.npz"), des=des, codes=codes, scipy_version=np.array(scipy.__version__))
print("wrote golden_pq.npz:", des.shape, codes.shape, "scipy", scipy.__version__)
