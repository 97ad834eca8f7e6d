"""MI355X-native MVXNet hot path behind the reference's ``modules`` import surface."""
