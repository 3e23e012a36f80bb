// yaml.h — the YAML subset racer-tracer's config and scene files use.
//
// The reference reads YAML through the `config 0.13.3` crate
// (racer-tracer/src/config.rs:216-225, scene/yml.rs:152-171), which is not in
// /root/reference; its accepted schema is pinned by the seven shipped scene
// files and racer-tracer/config.yml.  Supported here: block mappings by
// indentation, block sequences ("- "), flow sequences/mappings, plain and
// quoted scalars, comments, "---".  Mapping order is preserved.
#pragma once
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace rthost {

struct YamlNode {
    enum Kind { Null, Scalar, Map, Seq } kind = Null;
    std::string scalar;
    std::vector<std::pair<std::string, YamlNode>> map; // insertion order
    std::vector<YamlNode> seq;
    int line = 0;

    bool is_null() const { return kind == Null; }
    bool is_scalar() const { return kind == Scalar; }
    bool is_map() const { return kind == Map; }
    bool is_seq() const { return kind == Seq; }
    // Case-insensitive lookup: the `config` crate lower-cases keys.
    const YamlNode *find(const std::string &key) const;
};

// Throws TracerError(Configuration) with file name and line on malformed input.
YamlNode parse_yaml(const std::string &text, const std::string &file_name);
YamlNode parse_yaml_file(const std::string &path);

bool iequals(const std::string &a, const std::string &b);

} // namespace rthost
