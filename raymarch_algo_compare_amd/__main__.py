"""`python -m raymarch_algo_compare_amd ...` -- the CLI of main.cli()."""
import sys

from .main import cli

if __name__ == "__main__":
    sys.exit(cli())
