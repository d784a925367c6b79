"""ViT-L/16 224 (benchmark configuration 5's network) cut to depth 4, batch 2, from the IMPORTED reference.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_vit_l16.py

Same recipe as make_golden.py (its ``gen``); a file of its own so the other fixtures are not rewritten.  The full
depth of 24 is not needed to pin the kernels' shapes (embed 1024, 16 heads, hidden 4096) and keeps the CPU run short.
"""
from make_golden import gen

if __name__ == "__main__":
    gen("vit_l16", batch=2, seed=13, depth=4, keep_stages=False, emb_rows=[0, 1, 50, 99, 196], tag="vit_l16_d4")
