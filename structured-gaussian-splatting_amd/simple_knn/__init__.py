"""Drop-in for the reference's second native dependency `simple_knn` (import site:
scene/gaussian_model.py:20 `from simple_knn._C import distCUDA2`), backed by libgsrast.so."""
