"""Format-independent building blocks (headers, payloads, frames, readers)."""
