class DispersionError(Exception):
    """Raised where disba raises disba._exception.DispersionError: the root search of the
    fundamental mode left its admissible range (caught by the reference at
    src/inversion/model.py:134 and src/inversion/inversion.py:356)."""
