#!/usr/bin/env python3
"""Builds tests/golden/mnist_1fc.json from DATA files of the reference (run in the build container only):
  data/mnist/MNIST/raw/t10k-images-idx3-ubyte.gz   first images, quantised to 2 bits as int(float32(p)/255*3)
                                                   (notebooks/mnist_quant_fc_inference.ipynb)
  data/mnist/2bits_test_mnist_labels.csv           their labels
  weights/mnist/1_layer/fc1_weight_50epochs_bs4_clamp128.csv   784 x 10 integer weights (the matrix the analyst encrypts row-wise
                                                   after transposition, hhe_pktnn_examples.cpp:474)
Only numbers are copied (inputs and the plain integer result they imply); no reference code."""
import gzip, json, os, struct
import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NIMG = 3
with gzip.open(os.path.join(REF, "data/mnist/MNIST/raw/t10k-images-idx3-ubyte.gz"), "rb") as f:
    magic, n, rows, cols = struct.unpack(">IIII", f.read(16))
    assert magic == 2051 and rows == 28 and cols == 28
    raw = np.frombuffer(f.read(NIMG * 784), dtype=np.uint8).reshape(NIMG, 784)
pix = (raw.astype(np.float32) / np.float32(255) * np.float32(3)).astype(np.int64)   # int() truncation
labels = [int(l) for l in open(os.path.join(REF, "data/mnist/2bits_test_mnist_labels.csv")).read().split()[:NIMG]]
w = np.array([[int(v) for v in line.strip().rstrip(",").split(",")] for line in
              open(os.path.join(REF, "weights/mnist/1_layer/fc1_weight_50epochs_bs4_clamp128.csv")) if line.strip()], dtype=np.int64)
assert w.shape == (784, 10)
wt = w.T                                   # one row per output neuron
logits = pix @ wt.T                        # plain integer FC (no bias on the HHE path)
out = {"source": "tests/golden/make_mnist_fixture.py (reference data files: t10k images, 2-bit labels csv, fc1 weights csv)",
       "quantisation": "int(float32(p)/255*3)", "labels": labels,
       "pixels": pix.tolist(), "weights_rows": wt.tolist(), "plain_logits": logits.tolist(),
       "argmax": [int(np.argmax(r)) for r in logits]}
path = os.path.join(ROOT, "tests", "golden", "mnist_1fc.json")
json.dump(out, open(path, "w"), separators=(",", ":"))
print("wrote", path, os.path.getsize(path), "bytes; labels", labels, "argmax", out["argmax"], "max |logit|", int(np.abs(logits).max()))
