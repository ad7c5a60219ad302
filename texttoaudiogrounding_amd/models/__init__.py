"""Host-side mirror of the reference's ``models`` package for the hot path."""
