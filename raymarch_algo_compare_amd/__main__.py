from .main import cli

raise SystemExit(cli())
