"""MI355X-native hot path of SelfRecon."""
