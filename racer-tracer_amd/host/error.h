// error.h — TracerError with the reference's exit codes
// (racer-tracer/src/error.rs:3-97).  Thrown inside the host layer, turned
// into integer codes at the C ABI (never thrown across it).
#pragma once
#include <stdexcept>
#include <string>
#include "../../include/rt_abi.h"

namespace rthost {

class TracerError : public std::runtime_error {
  public:
    TracerError(int code, const std::string &msg) : std::runtime_error(msg), code_(code) {}
    int code() const { return code_; }

    // Constructors named after the enum variants of error.rs, same messages.
    static TracerError Configuration(const std::string &file, const std::string &why) {
        return TracerError(RT_ERR_CONFIGURATION, "Config Error (" + file + "): " + why);
    }
    static TracerError ArgumentParsingError(const std::string &why) {
        return TracerError(RT_ERR_ARGUMENT_PARSING, "Argument parsing Error: " + why);
    }
    static TracerError UnknownMaterial(const std::string &name) {
        return TracerError(RT_ERR_UNKNOWN_MATERIAL, "Unknown Material " + name + ".");
    }
    static TracerError ImageSave(const std::string &why) {
        return TracerError(RT_ERR_IMAGE_SAVE, "Image save error: " + why);
    }
    static TracerError SceneLoad(const std::string &why) {
        return TracerError(RT_ERR_SCENE_LOAD, "Scene failed to load: " + why);
    }
    static TracerError FailedToOpenImage(const std::string &path, const std::string &why) {
        return TracerError(RT_ERR_FAILED_TO_OPEN_IMAGE, "Failed to open image " + path + ": " + why);
    }
    static TracerError FailedToParse(const std::string &what, const std::string &why) {
        return TracerError(RT_ERR_FAILED_TO_PARSE, "Failed to parse \"" + what + "\" into a vector: " + why);
    }

  private:
    int code_;
};

} // namespace rthost
