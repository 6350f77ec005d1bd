"""Host utilities of the reference's ``vall_e.utils`` that sit next to the hot path."""
