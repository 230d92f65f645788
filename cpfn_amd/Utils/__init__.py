"""Host-side mirror of the reference's `Utils` functions that sit on the device path (patch merging)."""
