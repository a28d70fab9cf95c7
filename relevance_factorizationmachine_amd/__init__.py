"""MI355X-native FM/MF training path."""
