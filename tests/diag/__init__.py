"""Device-vs-oracle diagnostics (run by hand on a GPU box: `python -m tests.diag.<name>`).  They live
under tests/ because they call the oracle, which only test infrastructure may do."""
