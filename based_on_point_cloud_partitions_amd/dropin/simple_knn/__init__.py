"""Drop-in for the reference's `simple_knn` package (submodules/simple-knn)."""
