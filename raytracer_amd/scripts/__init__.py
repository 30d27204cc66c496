"""numpy-only mirror of the reference's ``scripts`` package (same module and class names)."""
