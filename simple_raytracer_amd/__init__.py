"""MI355X-native ray-trace core for simple_raytracer scenes."""
