#!/usr/bin/env python3
"""Writes tests/golden/corpus_blobs.npz: the resolved-IR blob (this build's own format, include/game_hip.h) of every
cinematic in the reference's .game corpus - examples/, presets/, game-compiler/examples/ with stdlib/ as the import
library - compiled here by this build's front end. The GPU box has no reference tree; the blobs let it render the whole
corpus and compare every frame with the oracle.

    python tests/golden/make_corpus_blobs.py        (needs /root/reference)
"""
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import conftest as T  # noqa: E402


def main():
    game = T.load_product()
    out, refused = {}, []
    for d in ("examples", "presets", "game-compiler/examples"):
        for path in sorted(glob.glob(os.path.join(T.REFERENCE, d, "*.game"))):
            try:
                prog = game.compile_file(path, lib_dirs=[os.path.join(T.REFERENCE, "stdlib")])
            except game.GameError as e:
                refused.append((path, str(e)))
                continue
            for i, cin in enumerate(prog.cinematics):
                out[f"{d}/{os.path.basename(path)}#{i}"] = np.frombuffer(cin.blob, dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "corpus_blobs.npz"), **out)
    print(f"{len(out)} cinematics, {sum(len(v) for v in out.values())} bytes of IR; refused: {len(refused)}")
    for path, why in refused:
        print("  ", os.path.relpath(path, T.REFERENCE), "-", why.splitlines()[0][:100])


if __name__ == "__main__":
    main()
