"""Host-side helpers that feed the staging pipeline (multi-file sequences)."""
