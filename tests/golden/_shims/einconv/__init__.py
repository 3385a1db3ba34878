"""Stand-in for the un-vendored `einconv` dependency (padding helper only)."""
