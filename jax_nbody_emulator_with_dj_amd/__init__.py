"""MI355X-native N-body emulator engine behind the reference's Python API (work in progress)."""
