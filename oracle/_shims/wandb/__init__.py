"""Import-only stand-in so that `import wandb` at the top of the reference's
metamorph_llama.py succeeds inside the golden-vector generator.  Nothing on the
hot path touches wandb; this directory is only ever put on sys.path by
oracle/gen_golden.py (test infrastructure, never by the product)."""
