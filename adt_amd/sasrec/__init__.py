"""Mirror of the reference's sasrec/ package surface (model.py, utils.py, main.py) on the HIP hot path."""
