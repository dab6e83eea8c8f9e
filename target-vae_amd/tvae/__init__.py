"""tvae: host side of the MI355X-native TARGET-VAE training hot path."""
