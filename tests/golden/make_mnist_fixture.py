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

# ---- MNIST-scale fixture: the first 64 test images, 2-bit pixels packed four per byte (little end first), labels and the plain
#      integer logits of the same 784 x 10 layer (weights: mnist_1fc.json "weights_rows").  The reference's acceptance test over
#      the test set is prediction == label and HHE logits == plain matmul (hhe_pktnn_examples.cpp:692-699, 861-862).
N64 = 64
with gzip.open(os.path.join(REF, "data/mnist/MNIST/raw/t10k-images-idx3-ubyte.gz"), "rb") as f:
    f.read(16)
    raw64 = np.frombuffer(f.read(N64 * 784), dtype=np.uint8).reshape(N64, 784)
pix64 = (raw64.astype(np.float32) / np.float32(255) * np.float32(3)).astype(np.int64)
assert (pix64[:NIMG] == pix).all() and pix64.max() <= 3
packed = (pix64.reshape(N64, 196, 4) << (2 * np.arange(4))).sum(axis=2).astype(np.uint8)
labels64 = [int(l) for l in open(os.path.join(REF, "data/mnist/2bits_test_mnist_labels.csv")).read().split()[:N64]]
with gzip.open(os.path.join(REF, "data/mnist/MNIST/raw/t10k-labels-idx1-ubyte.gz"), "rb") as f:
    f.read(8)
    assert labels64 == list(f.read(N64))          # the csv and the idx file agree
logits64 = pix64 @ wt.T
out64 = {"source": "tests/golden/make_mnist_fixture.py (reference data files: t10k images 0..63, 2-bit labels csv; weights in mnist_1fc.json)",
         "quantisation": "int(float32(p)/255*3), four 2-bit pixels per byte, pixel 4k+i in bits 2i..2i+1 of byte k",
         "pixels_2bit_hex": [bytes(r).hex() for r in packed], "labels": labels64, "plain_logits": logits64.tolist(),
         "argmax": [int(np.argmax(r)) for r in logits64]}
path64 = os.path.join(ROOT, "tests", "golden", "mnist_64.json")
json.dump(out64, open(path64, "w"), separators=(",", ":"))
print("wrote", path64, os.path.getsize(path64), "bytes; plain accuracy on these", sum(a == l for a, l in zip(out64["argmax"], labels64)), "/", N64,
      "max |logit|", int(np.abs(logits64).max()))
