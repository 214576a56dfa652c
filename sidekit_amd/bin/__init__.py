"""Command-line drivers mirroring ``sidekit/bin`` for the extraction + scoring path."""
